"""Text rendering of one board (reference ``render_mode`` "text" / "text_full", gobblet.py:299-429).
Host-side inspection code, not part of the hot path.  Format restated, not copied: the same
information (top pieces per square; with ``full`` all three levels), laid out as the reference
draws it -- positions run down the columns (board.py:14-17)."""
from __future__ import annotations

import numpy as np


def _sym(v: int) -> str:
    if v == 0:
        return " . "
    return f"{int(v):+d} "


def render_text(env_or_squares, full: bool = False) -> str:
    sq = getattr(getattr(env_or_squares, "board", None), "squares", env_or_squares)
    sq = np.asarray(sq).astype(int).reshape(3, 9)
    lines = []
    if hasattr(env_or_squares, "turn"):
        lines.append(f"TURN: {env_or_squares.turn}, AGENT: {env_or_squares.agent_selection}, "
                     f"ACTION: {env_or_squares.action}")
    levels = [("SMALL", 0), ("MEDIUM", 1), ("LARGE", 2)] if full else []
    top = np.zeros(9, int)
    for p in range(9):
        col = sq[:, p]
        top[p] = col[2] if col[2] else (col[1] if col[1] else col[0])
    blocks = [("TOP", top)] + [(name, sq[k]) for name, k in levels]
    for name, cells in blocks:
        lines.append(f"[{name}]")
        for r in range(3):  # displayed row r holds positions r, r+3, r+6
            lines.append("|".join(_sym(cells[r + 3 * c]) for c in range(3)))
    return "\n".join(lines)
