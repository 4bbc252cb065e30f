"""Text rendering of one board: the reference's ``render_mode`` "text" and "text_full"
(gobblet.py:299-429; sample output README.md:132-159).  Host-side inspection code, not part of the
hot path.  The layout is restated from the reference's format -- a header line, then 3x3 grids of
7-character cells whose squares run down the columns (board.py:14-17) -- and checked character for
character against output captured from the reference (tests/golden/render_text.json)."""
from __future__ import annotations

import numpy as np

_BLANK = " " * 7 + "|" + " " * 7 + "|" + " " * 7
_RULE = "_" * 7 + "|" + "_" * 7 + "|" + "_" * 7


def _sym_full(v) -> str:  # gobblet.py:299-305: signed piece number
    v = int(v)
    return "- " if v == 0 else (f"+{v}" if v > 0 else f"{v}")


def _sym_size(v) -> str:  # gobblet.py:307-313: signed piece size
    v = int(v)
    return "- " if v == 0 else (f"+{(v + 1) // 2}" if v > 0 else f"{v // 2}")


def _cells(c) -> str:
    return f"  {c[0]}   " + "|" + f"   {c[1]}  " + "|" + f"   {c[2]}  "


def _grid_rows(grids):
    """grids: list of 9-symbol lists; yields the 9 text lines of the grids side by side."""
    join = lambda parts: "  ".join(parts)  # noqa: E731
    for r in range(3):
        yield join([_BLANK] * len(grids))
        yield join([_cells([g[r], g[r + 3], g[r + 6]]) for g in grids])
        yield join([_RULE if r < 2 else _BLANK] * len(grids))


def render_text(env, full: bool = False) -> str:
    """The text the reference prints for `env` (a raw_env-like object with ``board``, ``turn``,
    ``agent_selection``, ``action``), without the trailing newline of the last print()."""
    pos = env.action % 9
    piece = (env.action // 9) + 1
    lines = []
    if not full:
        piece = (piece + 1) // 2
        lines.append(f"TURN: {env.turn}, AGENT: {env.agent_selection}, ACTION: {env.action}, "
                     f"POSITION: {pos}, PIECE: {piece}")
        lines.extend(_grid_rows([[_sym_size(v) for v in env.board.get_flatboard()]]))
    else:
        lines.append(f"TURN: {env.turn}, AGENT: {env.agent_selection}, ACTION: {env.action}, "
                     f"POSITION: {pos}, PIECE: {piece}")
        lines.append(" " * 9 + "SMALL" + " " * 9 + "  " + " " * 10 + "MED" + " " * 10 + "  " + " " * 9 + "LARGE"
                     + " " * 9 + "  ")
        sq = np.asarray(env.board.squares)
        lines.extend(_grid_rows([[_sym_full(v) for v in sq[9 * k:9 * k + 9]] for k in range(3)]))
    lines.append("")
    return "\n".join(lines)
