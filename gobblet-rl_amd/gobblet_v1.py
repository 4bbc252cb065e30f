"""``gobblet_v1`` -- the reference's single-environment AEC surface (gobblet_rl/gobblet_v1.py:1-3 re-exports
``env``, ``parallel_env``, ``raw_env`` of gobblet_rl/game/gobblet.py:110-123) over the HIP engine.

``raw_env`` keeps the reference's attribute names and turn logic (gobblet.py:123-290); its ``board`` is a
``Board`` with the reference's interface (board.py) whose methods run on the GPU through a 1-board
``BatchedBoard``.  This is the plumbing configuration (BASELINE.json configs[0]); the throughput path is
``BatchedGobblet``.  PettingZoo / gymnasium are used when importable; otherwise small structural stand-ins
with the same attributes are used (they are third-party to the reference too).
"""
from __future__ import annotations

import ctypes
import threading

import numpy as np
import torch

from . import _native as nat
from .board import BatchedBoard
from .greedy_policy import GreedyGobbletPolicy  # noqa: F401  (gobblet_v1.py:2 re-exports it)

try:  # pragma: no cover - not installed in the build image
    from pettingzoo import AECEnv as _AECBase
    from pettingzoo.utils import wrappers as _pz_wrappers
    _HAVE_PZ = True
except Exception:  # noqa: BLE001
    _HAVE_PZ = False

    class _AECBase:  # structural stand-in for pettingzoo.AECEnv (PettingZoo 1.22.3 semantics)
        metadata: dict = {}

        def __init__(self):
            pass

        @property
        def unwrapped(self):
            return self

        def _clear_rewards(self):
            for agent in self.rewards:
                self.rewards[agent] = 0

        def _accumulate_rewards(self):
            for agent, reward in self.rewards.items():
                self._cumulative_rewards[agent] += reward

        def _deads_step_first(self):
            _deads_order = [a for a in self.agents if (self.terminations[a] or self.truncations[a])]
            if _deads_order:
                self._skip_agent_selection = self.agent_selection
                self.agent_selection = _deads_order[0]
            return self.agent_selection

        def _was_dead_step(self, action):
            if action is not None:
                raise ValueError("when an agent is dead, the only valid action is None")
            agent = self.agent_selection
            assert self.terminations[agent] or self.truncations[agent], \
                "an agent that was not dead as attempted to be removed"
            del self.terminations[agent], self.truncations[agent], self.rewards[agent]
            del self._cumulative_rewards[agent], self.infos[agent]
            self.agents.remove(agent)
            _deads_order = [a for a in self.agents if (self.terminations[a] or self.truncations[a])]
            if _deads_order:
                if getattr(self, "_skip_agent_selection", None) is None:
                    self._skip_agent_selection = self.agent_selection
                self.agent_selection = _deads_order[0]
            else:
                if getattr(self, "_skip_agent_selection", None) is not None:
                    self.agent_selection = self._skip_agent_selection
                self._skip_agent_selection = None
            self._clear_rewards()

        def agent_iter(self, max_iter=2 ** 63):
            it = 0
            while self.agents and it < max_iter:
                it += 1
                yield self.agent_selection

        def last(self, observe=True):
            agent = self.agent_selection
            observation = self.observe(agent) if observe else None
            return (observation, self._cumulative_rewards[agent], self.terminations[agent],
                    self.truncations[agent], self.infos[agent])

        def close(self):
            pass

try:  # pragma: no cover
    from gymnasium import spaces as _spaces
except Exception:  # noqa: BLE001
    class _Discrete:
        def __init__(self, n):
            self.n, self.shape, self.dtype = int(n), (), np.int64

        def contains(self, x):
            return isinstance(x, (int, np.integer)) and 0 <= int(x) < self.n

        def sample(self, mask=None):
            if mask is not None:
                return int(np.random.choice(np.flatnonzero(mask)))
            return int(np.random.randint(self.n))

        def __repr__(self):
            return f"Discrete({self.n})"

    class _Box:
        def __init__(self, low, high, shape, dtype):
            self.low, self.high, self.shape, self.dtype = low, high, tuple(shape), dtype

        def contains(self, x):
            x = np.asarray(x)
            return x.shape == self.shape and x.dtype == self.dtype and bool(((x >= self.low) & (x <= self.high)).all())

        def __repr__(self):
            return f"Box({self.low}, {self.high}, {self.shape}, {np.dtype(self.dtype).name})"

    class _Dict:
        def __init__(self, d):
            self.spaces = dict(d)

        def __getitem__(self, k):
            return self.spaces[k]

        def contains(self, x):
            return isinstance(x, dict) and set(x) == set(self.spaces) and all(
                self.spaces[k].contains(v) for k, v in x.items())

        def __repr__(self):
            return f"Dict({self.spaces})"

    class _spaces:  # noqa: N801
        Discrete, Box, Dict = _Discrete, _Box, _Dict


class _HipBoardEngine:
    """The 1-board engine behind ``Board``: ONE ``gbl_board_eval`` launch per query.  The board (27 B), the move
    (agent, action) and the 432-byte record live in one pinned host block mapped into the device's address space
    (``gbl_pinned_alloc``): the kernel reads and writes it directly, so a ply is one launch and one stream
    synchronisation -- no separate copies.  Stateless between calls (the caller passes the position every time),
    so one engine per device serves every ``Board``."""

    _STATE, _ACTION, _AGENT, _BYTES = 448, 432, 436, 512  # offsets in the block; the record is at 0

    def __init__(self, device):
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise nat.GobbletHipError("gobblet_v1 needs a GPU device (there is no CPU fallback)")
        self._lib = nat.lib()
        host, dev = ctypes.c_void_p(), ctypes.c_void_p()
        with torch.cuda.device(self.device):
            nat.check(self._lib.gbl_pinned_alloc(self._BYTES, ctypes.byref(host), ctypes.byref(dev)), "gbl_pinned_alloc")
        self._host, self._dev = host.value, dev.value
        block = np.ctypeslib.as_array((ctypes.c_int8 * self._BYTES).from_address(self._host))
        self._record = block[:nat.REC_BYTES]
        self._state = block[self._STATE:self._STATE + 27]
        self._action = block[self._ACTION:self._ACTION + 4].view(np.int32)
        self._agent = block[self._AGENT:self._AGENT + 1]
        self._fields = {k: self._record[o:o + size] for k, (o, size) in nat.REC_FIELDS.items()}
        self._index = self.device.index if self.device.index is not None else torch.cuda.current_device()
        self._lock = threading.Lock()  # one pinned block per device: environments stepped from several threads take turns

    def __del__(self):
        try:
            self._lib.gbl_pinned_free(self._host)
        except Exception:  # noqa: BLE001  (interpreter shutdown)
            pass

    def evaluate(self, squares, agent_index=None, action=None) -> dict:
        """Everything the reference derives from ``squares`` (int8[27]) -- after Board.play_turn(agent_index,
        action) when a move is given -- as numpy arrays (copies)."""
        with self._lock:  # write the block, launch, wait, read the block: one caller at a time
            self._state[:] = squares
            base = self._dev
            other = torch.cuda.current_device() != self._index  # (a launch goes to the CURRENT device)
            if other:
                prev = torch.cuda.current_device()
                torch.cuda.set_device(self._index)
            try:
                stream = torch.cuda.current_stream(self.device)  # the caller's stream of THIS call
                if action is None:
                    rc = self._lib.gbl_board_eval(base + self._STATE, None, None, base, 1, stream.cuda_stream)
                else:
                    self._action[0], self._agent[0] = action, agent_index
                    rc = self._lib.gbl_board_eval(base + self._STATE, base + self._AGENT, base + self._ACTION, base, 1,
                                                  stream.cuda_stream)
            finally:
                if other:
                    torch.cuda.set_device(prev)
            nat.check(rc, "gbl_board_eval")
            stream.synchronize()
            return {k: v.copy() for k, v in self._fields.items()}


class _HostBoardEngine:
    """The same one-board engine on the HOST flavour of the ABI (``gbl_cpu_board_eval``, include/gobblet_cpu.h) -- what
    ``env(device="cpu")`` runs on: BASELINE config 1 ("1 env ... on CPU, no GPU").  A flavour the caller asks for, never a
    fallback of the HIP path."""

    def __init__(self, device):
        self.device = torch.device(device)
        self._lib = nat.lib_for(self.device)
        self._block = np.zeros(_HipBoardEngine._BYTES, np.int8)
        self._record = self._block[:nat.REC_BYTES]
        self._state = self._block[_HipBoardEngine._STATE:_HipBoardEngine._STATE + 27]
        self._action = self._block[_HipBoardEngine._ACTION:_HipBoardEngine._ACTION + 4].view(np.int32)
        self._agent = self._block[_HipBoardEngine._AGENT:_HipBoardEngine._AGENT + 1]
        self._fields = {k: self._record[o:o + size] for k, (o, size) in nat.REC_FIELDS.items()}
        self._lock = threading.Lock()

    def evaluate(self, squares, agent_index=None, action=None) -> dict:
        with self._lock:
            self._state[:] = squares
            base = self._block.ctypes.data
            if action is None:
                self._lib.gbl_board_eval(base + _HipBoardEngine._STATE, None, None, base, 1, None)
            else:
                self._action[0], self._agent[0] = action, agent_index
                self._lib.gbl_board_eval(base + _HipBoardEngine._STATE, base + _HipBoardEngine._AGENT,
                                         base + _HipBoardEngine._ACTION, base, 1, None)
            return {k: v.copy() for k, v in self._fields.items()}


_ENGINES: dict = {}


def _new_backend(device):
    """The engine of a ``Board``: the HIP library for a GPU device, the host flavour of the same ABI for "cpu" -- one per
    device.  (Module-level so that the host-logic tests can monkeypatch it; nothing in the package does.)"""
    key = str(torch.device(device))
    if key not in _ENGINES:
        _ENGINES[key] = _HipBoardEngine(device) if torch.device(device).type == "cuda" else _HostBoardEngine(device)
    return _ENGINES[key]


_LINES = [(0, 1, 2), (3, 4, 5), (6, 7, 8), (0, 3, 6), (1, 4, 7), (2, 5, 8), (0, 4, 8), (2, 4, 6)]  # board.py:135-153


class Board:
    """One Gobblet board with the reference ``Board`` interface (board.py:4-242), evaluated on the GPU.

    ``squares`` is a host numpy float64[27] exactly like the reference's (callers read it, assign whole
    arrays and single cells: greedy_policy.py:71, manual_policy.py:60,194-196).  Whenever a query finds
    that its content changed, EVERYTHING the reference derives from the position -- winner, flat board,
    covered cells, both agents' legal masks and observations -- is computed in ONE launch
    (``gbl_board_eval``) and later queries on the same position are answered from that.  ``play_turn`` is
    fused in front of the same launch and primes the cache for the new position, so one ply of the AEC loop
    costs one launch.  The engine is always the HIP library (``_new_backend``); there is no fallback and no
    public way to swap it.
    """

    def __init__(self, squares=None, device="cuda:0"):
        self.squares = np.zeros(27)           # board.py:33
        self.squares_preview = np.zeros(27)   # board.py:34
        if squares is not None:
            self.squares = np.array(squares, dtype=np.float64).reshape(27)
        self._engine = _new_backend(device)
        self._key, self._cache = None, None
        self.calculate_winners()

    def calculate_winners(self):  # board.py:135-153
        self.winning_combinations = list(_LINES)

    def setup(self):  # board.py:36-37
        self.calculate_winners()

    def _key_of(self):
        return np.ascontiguousarray(self.squares, dtype=np.float64).tobytes()

    def _position(self):
        sq = np.asarray(self.squares)
        if sq.shape != (27,) or not np.all(sq == np.round(sq)) or np.abs(sq).max(initial=0) > 6:
            raise ValueError("Board.squares must be 27 integers in [-6, 6]")
        return sq.astype(np.int8)

    def _evaluate(self):
        key = self._key_of()
        if key != self._key:
            self._cache = self._engine.evaluate(self._position())
            self._key = key
        return self._cache

    # decoders, board.py:42-79
    get_action_from_pos_piece = staticmethod(BatchedBoard.get_action_from_pos_piece)
    get_pos_from_action = staticmethod(BatchedBoard.get_pos_from_action)
    get_piece_from_action = staticmethod(BatchedBoard.get_piece_from_action)
    get_piece_size_from_action = BatchedBoard.get_piece_size_from_action
    get_index_from_action = BatchedBoard.get_index_from_action

    @staticmethod
    def _agent(agent_index):
        return 0 if agent_index == 0 else 1  # board.py:86: anything but 0 plays as player_2

    def is_legal(self, action, agent_index=0):  # board.py:82-115
        action = int(action)
        return 0 <= action < 54 and bool(self._evaluate()["mask%d" % self._agent(agent_index)][action])

    def play_turn(self, agent_index, action):  # board.py:118-132
        if agent_index not in (0, 1):
            raise ValueError("agent_index must be 0 or 1")
        action = int(action) if 0 <= int(action) < 54 else -1  # out of range = illegal = no-op
        # the move and the new position's record in one launch
        self._cache = self._engine.evaluate(self._position(), int(agent_index), action)
        self.squares = self._cache["squares"].astype(np.float64)
        self._key = self.squares.tobytes()

    def get_action(self, pos, piece_size, agent_index):  # board.py:50-60: the first legal of the size's two pieces
        for piece in (2 * int(piece_size) - 2, 2 * int(piece_size) - 1):
            action = int(pos) + 9 * piece
            if self.is_legal(action, agent_index):
                return action
        return -1

    def get_flatboard(self):  # board.py:159-177
        return self._evaluate()["flat"].astype(np.float64)

    def check_for_winner(self):  # board.py:183-194
        return int(self._evaluate()["winner"][0])

    def check_game_over(self):  # board.py:196-201
        return self.check_for_winner() in (1, -1)

    def check_covered(self):  # board.py:203-220
        return self._evaluate()["covered"].astype(np.float64)

    def legal_moves(self, agent_index):
        return np.flatnonzero(self._evaluate()["mask%d" % self._agent(agent_index)]).tolist()

    def legal_mask(self, agent_index):
        """int8[54]: [is_legal(a, agent_index) for a in range(54)] (gobblet.py:223-228, 211-213), a copy"""
        return self._evaluate()["mask%d" % self._agent(agent_index)].astype(np.int8)

    def observation(self, agent_index):
        return self._evaluate()["obs%d" % self._agent(agent_index)].reshape(3, 3, 13).copy()

    def print_pieces(self):  # board.py:223-239 (DEBUG helper, called by render() when args.debug is set)
        """Five lines describing the position, in the reference's wording and list formats.  One quirk is kept:
        "squares with uncovered pieces" tests ``check_covered()[pos]`` with the square number 0..8 as the index,
        i.e. it looks at the small-piece level only, whichever level the piece is on."""
        sq, covered = np.asarray(self.squares), self.check_covered()
        occupied = [c % 9 for c in range(27) if sq[c] != 0]
        print("open_indices: ", [c for c in range(27) if sq[c] == 0])
        print("open_squares: ", [np.where(self.get_flatboard() == 0)[0]])
        print("squares with pieces: ", occupied)
        print("squares with uncovered pieces: ", [q % 9 for q in occupied if covered[q] == 0])
        print("squares with covered pieces: ", [c % 9 for c in np.where(covered == 1)[0]])

    def print(self):  # board.py:155-156
        print(self.get_flatboard().reshape(3, 3).transpose())

    def __str__(self):  # board.py:241-242
        return str(np.asarray(self.squares).reshape(3, 3, 3))


class _AgentSelector:  # pettingzoo.utils.agent_selector (round robin)
    def __init__(self, agent_order):
        self.reinit(agent_order)

    def reinit(self, agent_order):
        self.agent_order = agent_order
        self._current_agent = 0
        self.selected_agent = 0

    def reset(self):
        self.reinit(self.agent_order)
        return self.next()

    def next(self):
        self._current_agent = (self._current_agent + 1) % len(self.agent_order)
        self.selected_agent = self.agent_order[self._current_agent - 1]
        return self.selected_agent


class raw_env(_AECBase):  # noqa: N801  (reference spelling, gobblet.py:123)
    metadata = {
        "render_modes": ["text", "text_full"],  # pygame modes ("human", "rgb_array") are out of scope
        "name": "gobblet_v1",
        "is_parallelizable": True,  # gobblet.py:127 (parallel_env below is PettingZoo's conversion of the AEC environment)
        "render_fps": 60,
        "has_manual_policy": False,
    }

    def __init__(self, render_mode=None, args=None, device="cuda:0"):
        super().__init__()
        self._device = device
        self.board = Board(device=device)
        self.board_size = 3
        self.agents = ["player_1", "player_2"]          # gobblet.py:137
        self.possible_agents = self.agents[:]
        self.action_spaces = {i: _spaces.Discrete(54) for i in self.agents}  # gobblet.py:140
        self.observation_spaces = {                      # gobblet.py:141-153
            i: _spaces.Dict({
                "observation": _spaces.Box(low=0, high=1, shape=(3, 3, 13), dtype=np.int8),
                "action_mask": _spaces.Box(low=0, high=1, shape=(54,), dtype=np.int8),
            }) for i in self.agents
        }
        self.rewards = {i: 0 for i in self.agents}
        self.terminations = {i: False for i in self.agents}
        self.truncations = {i: False for i in self.agents}
        self.infos = {i: {"legal_moves": list(range(0, 9))} for i in self.agents}  # gobblet.py:158
        self._agent_selector = _AgentSelector(self.agents)
        self.agent_selection = self._agent_selector.reset()
        self.render_mode = render_mode
        self.debug = args.debug if hasattr(args, "debug") else False
        self.screen_width = args.screen_width if hasattr(args, "screen_width") else 640  # gobblet.py:165-166 (pygame window;
        self.screen_height = self.screen_width                                           #  kept as attributes only)
        self.screen = None

    def observe(self, agent):  # gobblet.py:179-215
        idx = self.possible_agents.index(agent)
        observation = self.board.observation(idx)
        # gobblet.py:209-213: the legal moves of the agent to move as a 0/1 vector; all zeros for the other agent
        if agent == self.agent_selection:
            action_mask = self.board.legal_mask(idx)
        else:
            action_mask = np.zeros(54, "int8")
        return {"observation": observation, "action_mask": action_mask}

    def observation_space(self, agent):
        return self.observation_spaces[agent]

    def action_space(self, agent):
        return self.action_spaces[agent]

    def _legal_moves(self):  # gobblet.py:223-228
        return self.board.legal_moves(self.possible_agents.index(self.agent_selection))

    def step(self, action):  # gobblet.py:231-273
        if self.terminations[self.agent_selection] or self.truncations[self.agent_selection]:
            return self._was_dead_step(action)
        # gobblet.py:238-242.  The reference hands is_legal the agent NAME as agent_index, and Board.is_legal tests
        # `agent_index == 0` (board.py:86): whoever moves, this debug test is player_2's.
        if self.debug and not self.board.is_legal(action, self.agent_selection):
            print("piece: ", self.board.get_piece_from_action(action))
            print("piece_size: ", self.board.get_piece_size_from_action(action))
            print("pos: ", self.board.get_pos_from_action(action))
            print("--ERROR-- ILLEGAL MOVE")
        self.board.play_turn(self.agents.index(self.agent_selection), action)  # illegal: silent no-op
        next_agent = self._agent_selector.next()
        if self.board.check_game_over():
            winner = self.board.check_for_winner()
            if winner == 1:
                self.rewards[self.agents[0]] += 1
                self.rewards[self.agents[1]] -= 1
            elif winner == -1:
                self.rewards[self.agents[1]] += 1
                self.rewards[self.agents[0]] -= 1
            self.terminations = {i: True for i in self.agents}
        self._cumulative_rewards[self.agent_selection] = 0
        self.agent_selection = next_agent
        self._accumulate_rewards()
        self.turn += 1
        self.action = action
        if self.render_mode in ["text", "text_full"]:
            self.render()

    def reset(self, seed=None, return_info=False, options=None):  # gobblet.py:275-290 (seed is ignored)
        self.board = Board(device=self._device)
        self.agents = self.possible_agents[:]
        self.rewards = {i: 0 for i in self.agents}
        self._cumulative_rewards = {i: 0 for i in self.agents}
        self.terminations = {i: False for i in self.agents}
        self.truncations = {i: False for i in self.agents}
        self.infos = {i: {} for i in self.agents}
        self._agent_selector.reinit(self.agents)
        self._agent_selector.reset()
        self.agent_selection = self._agent_selector.reset()
        self.turn = 0
        self.action = -1

    def render(self):  # gobblet.py:292-429, text modes only
        if self.render_mode is None:
            import warnings
            warnings.warn("You are calling render method without specifying any render mode.")
            return
        from .render import render_text
        if self.debug:  # gobblet.py:315-316
            self.board.print_pieces()
        if self.render_mode == "text" or self.debug:  # gobblet.py:317
            full = False
        elif self.render_mode == "text_full":         # gobblet.py:342
            full = True
        else:
            raise NotImplementedError("pygame render modes ('human', 'rgb_array') are out of scope (SURVEY.md 8)")
        out = render_text(self, full=full)
        print(out)  # (the reference ends with an empty print(); `out` ends with that newline's line)
        return out

    def close(self):
        pass


class _EnvWrappers:
    """``env()`` of the reference stacks TerminateIllegalWrapper(illegal_reward=-1) ->
    AssertOutOfBoundsWrapper -> OrderEnforcingWrapper (gobblet.py:110-117).  Those wrappers are
    PettingZoo's; when PettingZoo is not importable this class restates what they do to this
    environment: an action outside Discrete(54) asserts; stepping / observing before reset raises;
    an action whose cached mask bit is 0 gives the mover -1, everyone else 0, and terminates and
    truncates every agent with the board untouched (gobblet.py:50-51)."""

    def __init__(self, raw):
        self.env = raw
        self._has_reset = False
        self._terminated = False
        self._prev_obs = None

    def __getattr__(self, name):
        if name.startswith("_") and name not in ("_cumulative_rewards", "_legal_moves"):
            raise AttributeError(name)
        return getattr(self.env, name)

    @property
    def unwrapped(self):
        return self.env

    def reset(self, seed=None, return_info=False, options=None):
        self._has_reset, self._terminated, self._prev_obs = True, False, None
        self.env.reset(seed=seed, options=options)

    def observe(self, agent):
        if not self._has_reset:
            raise AttributeError("reset() needs to be called before observe")
        obs = self.env.observe(agent)
        if agent == self.env.agent_selection:
            self._prev_obs = obs
        return obs

    def last(self, observe=True):
        agent = self.env.agent_selection
        observation = self.observe(agent) if observe else None
        e = self.env
        return (observation, e._cumulative_rewards[agent], e.terminations[agent], e.truncations[agent],
                e.infos[agent])

    def agent_iter(self, max_iter=2 ** 63):
        if not self._has_reset:
            raise AttributeError("reset() needs to be called before agent_iter")
        return self.env.agent_iter(max_iter)

    def step(self, action):
        if not self._has_reset:
            raise AttributeError("reset() needs to be called before step")
        e = self.env
        agent = e.agent_selection
        dead = e.terminations[agent] or e.truncations[agent]
        assert (action is None and dead) or e.action_space(agent).contains(action), \
            "action is not in action space"
        if self._prev_obs is None:
            self.observe(agent)
        prev_mask = self._prev_obs["action_mask"]
        self._prev_obs = None
        if self._terminated and dead:
            e._was_dead_step(action)
        elif not dead and not prev_mask[action]:
            e._cumulative_rewards[agent] = 0
            e.terminations = {d: True for d in e.agents}
            e.truncations = {d: True for d in e.agents}
            e.rewards = {d: 0 for d in e.truncations}
            e.rewards[agent] = float(-1)
            e._accumulate_rewards()
            e._deads_step_first()
            self._terminated = True
        else:
            e.step(action)

    def render(self):
        return self.env.render()

    def close(self):
        self.env.close()


def env(render_mode=None, args=None, device="cuda:0"):  # gobblet.py:110-117
    e = raw_env(render_mode=render_mode, args=args, device=device)
    if _HAVE_PZ:  # pragma: no cover
        e = _pz_wrappers.TerminateIllegalWrapper(e, illegal_reward=-1)
        e = _pz_wrappers.AssertOutOfBoundsWrapper(e)
        return _pz_wrappers.OrderEnforcingWrapper(e)
    return _EnvWrappers(e)


class _AecToParallel:
    """``parallel_env = parallel_wrapper_fn(env)`` (gobblet.py:120): PettingZoo's AEC -> parallel conversion, restated from
    pettingzoo.utils.conversions.aec_to_parallel_wrapper (1.22.3; third-party, not under the reference tree and not installed
    here -- PARITY UNPINNED, like the wrappers of ``env()``; upstream skips its own test of it, tests/test_gobblet_env.py:37-43).
    One ``step(actions)`` lets every live agent move once, in turn order, and sums the rewards of the cycle.  What the
    conversion does to a strictly turn-based game is kept as it is: when the first mover ends the game, the second agent's
    action is handed to an agent that is already dead and the AEC environment raises -- pass ``None`` for it."""

    def __init__(self, aec_env):
        assert aec_env.metadata.get("is_parallelizable", False), \
            "Converting from an AEC environment to a parallel environment with the to_parallel wrapper is not generally safe"
        self.aec_env = aec_env
        self.possible_agents = aec_env.possible_agents
        self.metadata = aec_env.metadata
        self.agents = []

    @property
    def unwrapped(self):
        return self.aec_env.unwrapped

    def observation_space(self, agent):
        return self.aec_env.observation_space(agent)

    def action_space(self, agent):
        return self.aec_env.action_space(agent)

    def reset(self, seed=None, return_info=False, options=None):
        e = self.aec_env
        e.reset(seed=seed, options=options)
        self.agents = e.agents[:]
        observations = {a: e.observe(a) for a in e.agents if not (e.terminations[a] or e.truncations[a])}
        return (observations, dict(**e.infos)) if return_info else observations

    def step(self, actions):
        e = self.aec_env
        rewards = {}
        for agent in list(e.agents):
            if agent != e.agent_selection:
                if e.terminations[agent] or e.truncations[agent]:
                    raise AssertionError(f"expected agent {agent} got termination or truncation agent {e.agent_selection}. "
                                         "Parallel environment wrapper expects all agent death to happen only at the end of a cycle.")
                raise AssertionError(f"expected agent {agent} got agent {e.agent_selection}, "
                                     "Parallel environment wrapper expects agents to step in a cycle.")
            e.last()
            e.step(actions[agent])
            for a in e.agents:
                rewards[a] = rewards.get(a, 0) + e.rewards[a]
        terminations, truncations, infos = dict(**e.terminations), dict(**e.truncations), dict(**e.infos)
        observations = {a: e.observe(a) for a in e.agents}
        while e.agents and (e.terminations[e.agent_selection] or e.truncations[e.agent_selection]):
            e.step(None)
        self.agents = e.agents
        return observations, rewards, terminations, truncations, infos

    def render(self):
        return self.aec_env.render()

    def close(self):
        return self.aec_env.close()


def parallel_env(render_mode=None, args=None, device="cuda:0"):  # gobblet.py:120
    if _HAVE_PZ:  # pragma: no cover
        from pettingzoo.utils.conversions import aec_to_parallel_wrapper
        return aec_to_parallel_wrapper(env(render_mode=render_mode, args=args, device=device))
    return _AecToParallel(env(render_mode=render_mode, args=args, device=device))
