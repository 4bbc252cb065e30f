"""``BatchedBoard`` -- the reference ``Board`` interface (gobblet_rl/game/board.py) over N boards in HBM.

Same method names, argument meaning and results as the reference class, but every call acts on
all N boards in lockstep through the HIP library and returns torch tensors on the device.  The
state is the attribute ``squares`` -- an int8 ``(N, 27)`` tensor whose row b is board b's
``Board.squares`` 27-vector (``squares[9*level + pos]``, board.py:6-33); like the reference's
attribute it may be read and assigned (whole tensor, rows or single cells).
"""
from __future__ import annotations

import torch

from . import _native as nat


def _as_i8(x, n, device, name):
    """scalar | sequence | tensor -> contiguous int8 device tensor of length n"""
    if isinstance(x, torch.Tensor):
        t = x.to(device=device, dtype=torch.int8)
        if t.dim() == 0:
            t = t.expand(n)
    else:
        t = torch.as_tensor(x, dtype=torch.int8, device=device)
        if t.dim() == 0:
            t = t.expand(n)
    if t.shape != (n,):
        raise ValueError(f"{name} must be a scalar or have shape ({n},), got {tuple(t.shape)}")
    return t.contiguous()


def _as_i32(x, n, device, name):
    t = x.to(device=device, dtype=torch.int32) if isinstance(x, torch.Tensor) else torch.as_tensor(
        x, dtype=torch.int32, device=device)
    if t.dim() == 0:
        t = t.expand(n)
    if t.shape != (n,):
        raise ValueError(f"{name} must be a scalar or have shape ({n},), got {tuple(t.shape)}")
    return t.contiguous()


class BatchedBoard:
    """N Gobblet boards on one MI355X.  Mirrors ``Board`` (board.py:4-242)."""

    def __init__(self, num_envs: int, device="cuda:0", squares: torch.Tensor | None = None):
        if num_envs < 1:
            raise ValueError("num_envs must be >= 1")
        self.device = torch.device(device)
        # "cuda": the HIP library (it raises if it cannot be built or loaded: no fallback); "cpu": the host flavour of the same
        # ABI (include/gobblet_cpu.h), which a caller has to ask for
        self.num_envs = int(num_envs)
        self._lib = nat.lib_for(self.device)
        # board.py:33: np.zeros(27)
        self._squares = torch.zeros((self.num_envs, nat.CELLS), dtype=torch.int8, device=self.device)
        if squares is not None:
            self.squares = squares
        self.calculate_winners()

    # -- state ---------------------------------------------------------------------------------
    @property
    def squares(self) -> torch.Tensor:
        return self._squares

    @squares.setter
    def squares(self, value):
        v = torch.as_tensor(value, device=self.device).to(torch.int8).reshape(self.num_envs, nat.CELLS)
        self._squares = v.contiguous().clone() if v.data_ptr() % 16 else v.contiguous()

    def _stream(self):
        return nat.current_stream(self.device)

    # -- board.py:135-153 ------------------------------------------------------------------------
    def calculate_winners(self):
        idx = list(range(9))
        combos = [tuple(idx[i:i + 3]) for i in range(0, 9, 3)]
        combos += [tuple(idx[x] for x in range(y, 9, 3)) for y in range(3)]
        combos.append(tuple(range(0, 9, 4)))
        combos.append(tuple(range(2, 8, 2)))
        self.winning_combinations = combos

    def setup(self):
        self.calculate_winners()

    # -- action decoders, board.py:42-79 (elementwise on ints or tensors) -------------------------
    @staticmethod
    def get_action_from_pos_piece(pos, piece):
        return 9 * (piece - 1) + pos if (pos in range(9) and piece in range(1, 7)) else -1

    @staticmethod
    def get_pos_from_action(action):
        return action % 9

    @staticmethod
    def get_piece_from_action(action):
        return (action // 9) + 1

    @classmethod
    def get_piece_size_from_action(cls, action):
        return (cls.get_piece_from_action(action) + 1) // 2

    @classmethod
    def get_index_from_action(cls, action):
        return cls.get_pos_from_action(action) + 9 * (cls.get_piece_size_from_action(action) - 1)

    # -- board.py:82-115 -----------------------------------------------------------------------------
    def is_legal(self, action, agent_index=0) -> torch.Tensor:
        """bool (N,): is ``action[b]`` legal for ``agent_index[b]`` on board b (scalars broadcast)."""
        n = self.num_envs
        a = _as_i32(action, n, self.device, "action")
        ag = _as_i8(agent_index, n, self.device, "agent_index")
        out = torch.empty(n, dtype=torch.int8, device=self.device)
        nat.check(self._lib.gbl_is_legal(self._squares.data_ptr(), ag.data_ptr(), a.data_ptr(), out.data_ptr(), n,
                                         self._stream()), "gbl_is_legal")
        return out.bool()

    def legal_mask(self, agent_index) -> torch.Tensor:
        """int8 (N, 54): ``[is_legal(a, agent_index) for a in range(54)]`` (gobblet.py:223-228, 211-213)."""
        n = self.num_envs
        ag = _as_i8(agent_index, n, self.device, "agent_index")
        out = torch.empty((n, nat.ACTIONS), dtype=torch.int8, device=self.device)
        nat.check(self._lib.gbl_legal_mask(self._squares.data_ptr(), ag.data_ptr(), out.data_ptr(), n, self._stream()),
                  "gbl_legal_mask")
        return out

    # -- board.py:118-132 -----------------------------------------------------------------------------
    def play_turn(self, agent_index, action) -> None:
        n = self.num_envs
        a = _as_i32(action, n, self.device, "action")
        ag = _as_i8(agent_index, n, self.device, "agent_index")
        nat.check(self._lib.gbl_play_turn(self._squares.data_ptr(), ag.data_ptr(), a.data_ptr(), n, self._stream()),
                  "gbl_play_turn")

    # -- board.py:50-60 ---------------------------------------------------------------------------------
    def get_action(self, pos, piece_size, agent_index) -> torch.Tensor:
        """int32 (N,): first legal of the two pieces of ``piece_size`` at ``pos``, else -1."""
        n = self.num_envs
        pos_t = _as_i32(pos, n, self.device, "pos")
        size_t = _as_i32(piece_size, n, self.device, "piece_size")
        a1 = pos_t + 9 * (size_t * 2 - 2)
        a2 = pos_t + 9 * (size_t * 2 - 1)
        l1 = self.is_legal(a1, agent_index)
        l2 = self.is_legal(a2, agent_index)
        return torch.where(l1, a1, torch.where(l2, a2, torch.full_like(a1, -1)))

    # -- board.py:159-177 -------------------------------------------------------------------------------
    def get_flatboard(self) -> torch.Tensor:
        n = self.num_envs
        out = torch.empty((n, 9), dtype=torch.int8, device=self.device)
        nat.check(self._lib.gbl_flatboard(self._squares.data_ptr(), out.data_ptr(), n, self._stream()), "gbl_flatboard")
        return out

    # -- board.py:183-201 -------------------------------------------------------------------------------
    def check_for_winner(self) -> torch.Tensor:
        n = self.num_envs
        out = torch.empty(n, dtype=torch.int8, device=self.device)
        nat.check(self._lib.gbl_winner(self._squares.data_ptr(), out.data_ptr(), n, self._stream()), "gbl_winner")
        return out

    def check_game_over(self) -> torch.Tensor:
        return self.check_for_winner() != 0

    # -- board.py:203-220 -------------------------------------------------------------------------------
    def check_covered(self) -> torch.Tensor:
        n = self.num_envs
        out = torch.empty((n, nat.CELLS), dtype=torch.int8, device=self.device)
        nat.check(self._lib.gbl_covered(self._squares.data_ptr(), out.data_ptr(), n, self._stream()), "gbl_covered")
        return out

    # -- everything the reference derives from a position, in one launch ---------------------------------
    def evaluate(self, agent_index=None, action=None, out: torch.Tensor | None = None) -> dict:
        """``gbl_board_eval``: optionally ``play_turn(agent_index, action)`` first, then per board the winner,
        flat board, covered cells, both agents' legal masks and both observations -- one launch instead of seven.
        Returns views into one int8 ``(N, 432)`` record tensor (``"record"``) keyed "squares", "winner", "flat",
        "covered", "mask0", "mask1", "obs0", "obs1"."""
        n = self.num_envs
        rec = out if out is not None else torch.empty((n, nat.REC_BYTES), dtype=torch.int8, device=self.device)
        a = ag = None
        if action is not None:
            a = _as_i32(action, n, self.device, "action")
            ag = _as_i8(agent_index, n, self.device, "agent_index")
        nat.check(self._lib.gbl_board_eval(self._squares.data_ptr(), nat.ptr(ag), nat.ptr(a), rec.data_ptr(), n,
                                           self._stream()), "gbl_board_eval")
        views = {k: rec[:, o:o + size] for k, (o, size) in nat.REC_FIELDS.items()}
        views["winner"] = views["winner"][:, 0]
        views["obs0"], views["obs1"] = views["obs0"].reshape(n, 3, 3, 13), views["obs1"].reshape(n, 3, 3, 13)
        views["record"] = rec
        return views

    # -- state contract (for callers that assign ``squares``) --------------------------------------------
    def validate(self, raise_on_error: bool = True) -> torch.Tensor:
        """int8 (N,) flags: bit 0 = a cell holds a value its level cannot hold, bit 1 = a piece number
        occurs twice.  With ``raise_on_error`` a duplicate raises what the reference's ``is_legal`` raises
        (board.py:94-95) and an impossible value raises ValueError."""
        n = self.num_envs
        out = torch.empty(n, dtype=torch.int8, device=self.device)
        nat.check(self._lib.gbl_validate(self._squares.data_ptr(), out.data_ptr(), n, self._stream()), "gbl_validate")
        if raise_on_error:
            worst = int(out.max())
            if worst & 2:
                raise Exception("PIECE HAS BEEN USED TWICE")
            if worst & 1:
                raise ValueError("a cell holds a value its level cannot hold (level k: 0, +-(2k+1), +-(2k+2))")
        return out

    # -- gobblet.py:179-208 (the observation planes are a pure function of the board) -------------------
    def observation(self, agent_index) -> torch.Tensor:
        """int8 (N, 3, 3, 13) as seen by ``agent_index`` (scalar 0/1 or an (N,) tensor)."""
        n = self.num_envs
        out = torch.empty((n, 3, 3, 13), dtype=torch.int8, device=self.device)
        if isinstance(agent_index, int):
            sel, tm = int(agent_index != 0), None
        else:
            sel, tm = -1, _as_i8(agent_index, n, self.device, "agent_index")
        nat.check(self._lib.gbl_observe(self._squares.data_ptr(), nat.ptr(tm), sel, out.data_ptr(), n, self._stream()),
                  "gbl_observe")
        return out

    def __str__(self):  # board.py:241-242
        return str(self._squares.reshape(self.num_envs, 3, 3, 3).cpu().numpy())
