"""``BatchedGobblet`` -- lockstep vectorised Gobblet over N boards on one MI355X.

One ``step(actions)`` is one ``raw_env.step`` (gobblet.py:231-271) plus the ``observe`` of the
agent that moves next (gobblet.py:179-215) on every board, fused in one HIP kernel
(``gbl_step``).  All state and all outputs live in HBM as torch tensors; nothing comes back to
the host unless the caller asks.

Tensors (attributes, all on ``device``)
    squares  int8 (N, 27)        Board.squares of every board
    turn     int32 (N,)          raw_env.turn per board (only with track_turn=True)
    to_move  int8 (N,)           index of agent_selection (0 = player_1)
    done     int8 (N,)           terminations (auto_reset: "episode ended on the last step")
    winner   int8 (N,)           check_for_winner() after the last step
    rewards  int8 (N, 2)         rewards of (player_1, player_2) for the last step
    action_mask int8 (N, 54)     legal mask of the agent to move
    observation int8 (N,3,3,13)  observation of the agent to move
"""
from __future__ import annotations

import torch

from . import _native as nat
from . import placement as _placement
from .board import BatchedBoard, _as_i32


class BatchedGobblet:
    SLOT_PAD_BOARDS = 0  # see trajectory_buffers
    STAGING_SETS = 2     # collect()'s own buffer sets kept at most (least recently used dropped first)

    metadata = {"name": "gobblet_v1_batched", "num_actions": nat.ACTIONS, "observation_shape": (3, 3, 13)}

    def __init__(self, num_envs: int, device="cuda:0", illegal_mode: str | int = "noop", auto_reset: bool = False,
                 with_observation: bool = True, seed: int = 0, env_base: int = 0, track_turn: bool = False):
        """illegal_mode: "noop" = raw_env semantics (silent no-op, the turn passes);
        "terminate" = env() semantics (TerminateIllegalWrapper: mover -1, episode ends).
        env_base: global index of this shard's board 0 (keys the sampler so results do not
        depend on how boards are sharded over GPUs).  track_turn: also keep ``turn`` (int32 (N,)),
        the reference's per-environment ``raw_env.turn`` (plies since that board's reset)."""
        self.board = BatchedBoard(num_envs, device)
        self.device = self.board.device
        self.num_envs = self.board.num_envs
        self._lib = nat.lib_for(self.device)
        modes = {"noop": nat.ILLEGAL_NOOP, "terminate": nat.ILLEGAL_TERMINATE, 0: 0, 1: 1}
        if illegal_mode not in modes:
            raise ValueError("illegal_mode must be 'noop' or 'terminate'")
        self.illegal_mode = modes[illegal_mode]
        self.auto_reset = bool(auto_reset)
        self.seed, self.env_base = int(seed), int(env_base)
        n, dev = self.num_envs, self.device
        self.to_move = torch.zeros(n, dtype=torch.int8, device=dev)
        self.done = torch.zeros(n, dtype=torch.int8, device=dev)
        self.winner = torch.zeros(n, dtype=torch.int8, device=dev)
        self.rewards = torch.zeros((n, 2), dtype=torch.int8, device=dev)
        self.action_mask = torch.empty((n, nat.ACTIONS), dtype=torch.int8, device=dev)
        self.observation = torch.empty((n, 3, 3, 13), dtype=torch.int8, device=dev) if with_observation else None
        self.actions = torch.zeros(n, dtype=torch.int32, device=dev)
        self.turn = torch.zeros(n, dtype=torch.int32, device=dev) if track_turn else None
        # rollout tallies, striped (include/gobblet_hip.h); totals via the ``counters`` property
        self._counters = torch.zeros((nat.COUNTER_STRIPES, nat.COUNTER_STRIDE), dtype=torch.int64, device=dev)
        self._ply, self._ply_dev = 0, None  # lockstep ply counter (keys the sampler), see ``ply`` / ``device_ply``
        self.policy_hist = None             # int8 (N, 2, 3): the device-side greedy policies' last three actions per agent
        self._staging = {}                  # collect()'s own trajectory buffers, one placed set per (plies, layout, outputs)
        self.reset()

    def reset_policy_history(self) -> None:
        """Empty action histories (-1) for the device-side greedy policies of ``collect(policies=...)``."""
        if self.policy_hist is None:
            self.policy_hist = torch.empty((self.num_envs, 2, 3), dtype=torch.int8, device=self.device)
        self.policy_hist.fill_(-1)

    @property
    def squares(self) -> torch.Tensor:
        return self.board.squares

    # -- the sampler's ply index ----------------------------------------------------------------------
    @property
    def ply(self) -> int:
        """Plies played in lockstep since reset; keys the masked-random sampler together with seed and board id."""
        return self._ply + (int(self._ply_dev.item()) if self._ply_dev is not None else 0)

    @ply.setter
    def ply(self, value: int) -> None:
        if self._ply_dev is not None:
            self._ply_dev.fill_(int(value))
            self._ply = 0
        else:
            self._ply = int(value)

    def device_ply(self, enable: bool = True) -> None:
        """Keep the ply index in device memory so that a captured hipGraph draws fresh random numbers on
        every replay: a kernel then uses (plies enqueued since the last ``advance_ply()``) + a device-resident
        base, and ``advance_ply()`` -- the last call of the captured sequence -- adds the former to the latter::

            env.device_ply()
            with torch.cuda.graph(g):
                for _ in range(K):
                    env.rollout(1)
                env.advance_ply()
            g.replay(); g.replay()   # 2K different plies

        Eager calls work unchanged in this mode."""
        if enable and self._ply_dev is None:
            self._ply_dev = torch.full((1,), self._ply, dtype=torch.int32, device=self.device)
            self._ply = 0
        elif not enable and self._ply_dev is not None:
            self._ply, self._ply_dev = self.ply, None

    def advance_ply(self) -> None:
        if self._ply_dev is not None and self._ply:
            nat.check(self._lib.gbl_counter_add(self._ply_dev.data_ptr(), self._ply, self._stream()), "gbl_counter_add")
            self._ply = 0

    @property
    def counters(self) -> torch.Tensor:
        """int64 (4,): plies played, games finished, player_1 wins, player_2 wins over all
        ``rollout(..., count=True)`` calls."""
        return self._counters.sum(0)[:4]

    def _stream(self):
        return nat.current_stream(self.device)

    # -- gobblet.py:275-290 --------------------------------------------------------------------------
    def reset(self, seed=None, options=None):
        """All boards empty, player_1 to move.  Like the reference, ``seed`` does not affect the
        (deterministic) environment; if given it re-keys the action sampler."""
        if seed is not None:
            self.seed = int(seed)
        n = self.num_envs
        nat.check(self._lib.gbl_reset(self.squares.data_ptr(), self.to_move.data_ptr(), self.done.data_ptr(),
                                      self.winner.data_ptr(), n, self._stream()), "gbl_reset")
        self.rewards.zero_()
        if self.turn is not None:
            self.turn.zero_()
        self.ply = 0
        # (collect()'s staging buffers are KEPT: a captured hipGraph may have their addresses baked in, and a reset-then-collect
        #  loop must not re-run the placement search every episode -- release_staging() is the explicit way to drop them)
        self.refresh()
        return self.observe()

    def refresh(self):
        """Recompute action_mask / observation from squares + to_move (after assigning state)."""
        n = self.num_envs
        nat.check(self._lib.gbl_legal_mask(self.squares.data_ptr(), self.to_move.data_ptr(),
                                           self.action_mask.data_ptr(), n, self._stream()), "gbl_legal_mask")
        if not self.auto_reset:
            self.action_mask.mul_((self.done == 0).to(torch.int8)[:, None])
        if self.observation is not None:
            nat.check(self._lib.gbl_observe(self.squares.data_ptr(), self.to_move.data_ptr(), -1,
                                            self.observation.data_ptr(), n, self._stream()), "gbl_observe")

    def reset_where(self, which) -> None:
        """Reset the boards selected by the bool mask `which` (N,) -- e.g. ``env.turn >= max_plies`` as a
        truncation guard; the reference itself never truncates (gobblet.py:250-252).  Host-side helper
        (torch indexing + refresh), not a hot-path kernel."""
        which = torch.as_tensor(which, device=self.device).bool()
        self.squares[which] = 0
        self.to_move[which] = 0
        self.done[which] = 0
        self.winner[which] = 0
        self.rewards[which] = 0
        if self.turn is not None:
            self.turn[which] = 0
        self.refresh()

    def observe(self):
        """{"observation", "action_mask"} of the agent to move on every board (gobblet.py:215)."""
        return {"observation": self.observation, "action_mask": self.action_mask}

    # -- checkpoint / resume: the whole environment is a handful of tensors ------------------------------
    def state_dict(self) -> dict:
        sd = {"squares": self.squares.clone(), "to_move": self.to_move.clone(), "done": self.done.clone(),
              "winner": self.winner.clone(), "rewards": self.rewards.clone(), "counters": self._counters.clone(),
              "ply": self.ply, "seed": self.seed, "env_base": self.env_base}
        if self.turn is not None:
            sd["turn"] = self.turn.clone()
        if self.policy_hist is not None:
            sd["policy_hist"] = self.policy_hist.clone()
        return sd

    def load_state_dict(self, sd: dict) -> None:
        """Copies INTO this environment's tensors: their addresses do not change (a hipGraph captured before the
        load stays valid) and the dict is not aliased (it can be loaded again, e.g. to rewind)."""
        n = self.num_envs
        self.squares.copy_(torch.as_tensor(sd["squares"], device=self.device).to(torch.int8).reshape(n, nat.CELLS))
        self.to_move.copy_(sd["to_move"]); self.done.copy_(sd["done"]); self.winner.copy_(sd["winner"])
        self.rewards.copy_(sd["rewards"]); self._counters.copy_(sd["counters"])
        if self.turn is not None:
            if "turn" in sd:
                self.turn.copy_(sd["turn"])
            else:
                self.turn.zero_()
        if "policy_hist" in sd:
            self.reset_policy_history()
            self.policy_hist.copy_(sd["policy_hist"])
        elif self.policy_hist is not None:
            self.reset_policy_history()
        self.ply, self.seed, self.env_base = int(sd["ply"]), int(sd["seed"]), int(sd["env_base"])
        self.refresh()

    # -- batches shaped like the reference's callers: zero-copy device views ---------------
    def tianshou_batch(self):
        """What Tianshou's PettingZooEnv wrapper hands a policy (greedy_policy_tianshou.py:63-84 reads
        ``obs.obs``, ``obs.mask``, ``agent_id``), for all boards: {"obs", "mask", "agent_id"}."""
        return {"obs": self.observation, "mask": self.action_mask.bool(), "agent_id": self.to_move}

    def rllib_batch(self):
        """The ``obs_batch`` of RLlib-style callers (greedy_policy.py:21-31): flat observation + mask."""
        return {"observation": self.observation.reshape(self.num_envs, -1), "action_mask": self.action_mask}

    # -- gobblet.py:231-271 + 179-215 ------------------------------------------------------------------
    def _i8_out(self, t, name):
        if t is None:
            return None
        if t.dtype != torch.int8 or t.device != self.device or t.numel() != self.num_envs or not t.is_contiguous():
            raise ValueError("%s: a contiguous int8 (N,) tensor on the environment's device" % name)
        return t

    def _i32_out(self, t, name):
        if t is None:
            return None
        if t.dtype != torch.int32 or t.device != self.device or t.numel() != self.num_envs or not t.is_contiguous():
            raise ValueError("%s: a contiguous int32 (N,) tensor on the environment's device" % name)
        return t

    def step(self, actions, status=None, next_actions=None):
        """Apply ``actions`` (int (N,)) for the agents to move.  Returns
        (obs dict, rewards (N,2), done (N,), winner (N,)) -- views of the attribute tensors.

        ``status`` (int8 (N,), optional): receives what became of every action -- 0 played, 1 (``nat.STATUS_ILLEGAL``) not a
        legal move of the mover (handled per ``illegal_mode``), 3 (| ``nat.STATUS_OUT_OF_RANGE``) outside [0, 54), where the
        reference's ``env()`` asserts (gobblet.py:110-117); a frozen board consumes no action: 0.
        ``next_actions`` (int32 (N,), optional; may be the ``actions`` tensor itself): receives the NEXT mover's masked-uniform
        draw from the mask this step stores -- what ``sample_actions()`` would return after the step, without its launch
        (``gbl_step_ex``): the random opponent's reply or an epsilon-greedy policy's exploration move."""
        n = self.num_envs
        a = _as_i32(actions, n, self.device, "actions")
        status, next_actions = self._i8_out(status, "status"), self._i32_out(next_actions, "next_actions")
        if status is None and next_actions is None:
            nat.check(self._lib.gbl_step(self.squares.data_ptr(), self.to_move.data_ptr(), self.done.data_ptr(),
                                         a.data_ptr(), self.winner.data_ptr(), self.rewards.data_ptr(),
                                         self.action_mask.data_ptr(), nat.ptr(self.observation), nat.ptr(self.turn), n,
                                         self.illegal_mode, int(self.auto_reset), self._stream()), "gbl_step")
        else:
            nat.check(self._lib.gbl_step_ex(self.squares.data_ptr(), self.to_move.data_ptr(), self.done.data_ptr(),
                                            a.data_ptr(), self.winner.data_ptr(), self.rewards.data_ptr(),
                                            self.action_mask.data_ptr(), nat.ptr(self.observation), nat.ptr(self.turn),
                                            None, None, None, nat.ptr(status), nat.ptr(next_actions), self.seed, self.env_base,
                                            self._ply + 1, nat.ptr(self._ply_dev), n, self.illegal_mode,
                                            int(self.auto_reset), self._stream()), "gbl_step_ex")
        self._ply += 1
        return self.observe(), self.rewards, self.done, self.winner

    # -- trajectory collection: T plies per launch, every ply materialised -------------------------------------
    POLICIES = {"random": nat.POLICY_RANDOM, "greedy1": nat.POLICY_GREEDY1, "greedy": nat.POLICY_GREEDY2,
                "greedy2": nat.POLICY_GREEDY2, "greedy3": nat.POLICY_GREEDY3}

    def trajectory_buffers(self, plies: int, layout: str = "time", pad_boards: int | None = None,
                           placement: str = "auto", policy_outputs: bool = False, candidates: bool = False,
                           far: bool | None = None) -> dict:
        """Device tensors for ``collect``.

        policy_outputs: also "chosen" (int32) and "how" (int8: 0 random ply, 1 greedy choice, 2 greedy fallback draw) for
        ``collect(policies=...)``; candidates: also "candidates" (int8 (..., 54): the greedy policy's ``actions_depth1``).

        placement "auto" (default): when the observation and the mask trajectory are large enough to be HBM streams
        (64 MiB each), the mask array is placed so that the two do not share one of the three 96 GiB classes of the
        device's memory, in which their writes would not overlap (``placement.py``: a probe kernel measures the pair as
        torch's allocator places it; only if that pair shares a class either array becomes the head of a hipMalloc block
        of its own of at least 2 GiB -- a 64 MiB mask array then pins 2 GiB for its lifetime -- and a few more blocks are
        held while searching: at most 64 GiB and at most a quarter of what is free, all handed straight back to the driver
        afterwards, torch's cache is never flushed; the best pair seen wins, so the result is never worse than the
        allocator's own; 33 -> 27 us per ply at 2^20 boards).  When the device cannot spare the blocks the arrays are plain torch allocations and ``_placement``
        says why.  "spread" insists (raises if the arrays are too small to probe or the memory is not there), "any" takes
        the allocator's addresses as they come.  The search synchronises the device and takes a few milliseconds: make
        the buffers once and reuse them (``collect(out=...)``).  What happened is recorded under ``_placement``
        (``ratio`` >= 0.95: the arrays share a class -- "unplaced").  ``far``: see ``placement.spread_pair`` (True = a caller
        that owns the device lets the search reach behind transient gaps of 64 - 128 GiB; False = never; None = only on a
        device with 160 GiB free).

        layout "time" (default): every entry has shape (plies, N, ...) -- one slice per ply, a view of a
        (plies, slot_boards, ...) allocation; slot_boards = N rounded up to 128 boards (+ ``pad_boards``), so that
        every slot of every array starts on a 128-byte line (the C-ABI itself asks for multiples of 16 only).
        layout "tile": every entry has shape (tiles, plies, 64, ...), tiles = ceil(N / 64) -- board b is
        [b // 64, :, b % 64]; each tile keeps its whole trajectory contiguous, so every wavefront of the kernel writes
        one sequential region per array (boards past N in the last tile are never written)."""
        n, dev, T = self.num_envs, self.device, int(plies)
        if layout not in ("time", "tile"):
            raise ValueError("layout must be 'time' or 'tile'")
        tiles = -(-n // 64)
        if layout == "time":
            slot = -(-n // 128) * 128 + (self.SLOT_PAD_BOARDS if pad_boards is None else int(pad_boards))
            lead, ply_stride, tile_stride = (T, slot), slot, 64
        else:
            lead, ply_stride, tile_stride = (tiles, T, 64), 64, 64 * T
        if placement not in ("auto", "spread", "any"):
            raise ValueError("placement must be 'auto', 'spread' or 'any'")
        full, placed = {}, {"spread": False, "why": "placement='any'"}

        def make_obs():
            return torch.zeros(lead + (3, 3, 13), dtype=torch.int8, device=dev)

        def make_mask():
            return torch.zeros(lead + (nat.ACTIONS,), dtype=torch.int8, device=dev)

        cells = T * ply_stride if layout == "time" else tiles * T * 64
        probeable = dev.type == "cuda" and self.observation is not None and cells * nat.ACTIONS >= _placement.MIN_BYTES
        if placement == "spread" and not probeable:
            raise ValueError("placement='spread' needs an observation trajectory and at least 64 MiB of mask trajectory")
        capturing = dev.type == "cuda" and torch.cuda.is_current_stream_capturing()
        if capturing and placement == "spread":
            raise ValueError("placement='spread' probes and synchronises: not inside a graph capture")
        if placement != "any" and probeable and capturing:
            # (the probe synchronises, which would break the capture: buffers made inside one lie as allocated)
            full["observation"], full["action_mask"] = make_obs(), make_mask()
            placed["why"] = "inside a graph capture"
        elif placement != "any" and probeable:
            geometry = dict(slot_boards=ply_stride, plies=T) if layout == "time" and ply_stride % 128 == 0 else {}
            cells = T * ply_stride if layout == "time" else tiles * T * 64
            try:
                a, b, placed = _placement.spread_pair(
                    cells * 117, cells * nat.ACTIONS, dev, **geometry, far=far,
                    plain=lambda: (torch.empty(cells * 117, dtype=torch.uint8, device=dev),
                                   torch.empty(cells * nat.ACTIONS, dtype=torch.uint8, device=dev)))
                full["observation"] = a.view(torch.int8).view(lead + (3, 3, 13))
                full["action_mask"] = b.view(torch.int8).view(lead + (nat.ACTIONS,))
            except _placement.PlacementUnavailable as e:
                # a device that is nearly full (a trainer's model and replay buffer): no search, the allocator's addresses
                if placement == "spread":
                    raise
                full["observation"], full["action_mask"] = make_obs(), make_mask()
                placed = {"spread": False, "why": "fell back to plain allocations: %s" % e}
        else:
            if self.observation is not None:
                full["observation"] = make_obs()
            full["action_mask"] = make_mask()
            if placement != "any":
                placed["why"] = ("host memory" if dev.type != "cuda" else
                                 "arrays too small to probe" if self.observation is not None else "no observation stream")
        extra = ((("chosen", torch.int32, ()), ("how", torch.int8, ())) if policy_outputs else ()) + \
                ((("candidates", torch.int8, (nat.ACTIONS,)),) if candidates else ())
        for key, dtype, tail in (("actions", torch.int32, ()), ("winner", torch.int8, ()), ("rewards", torch.int8, (2,)),
                                 ("done", torch.int8, ()), ("to_move", torch.int8, ())) + extra:
            full[key] = torch.zeros(lead + tail, dtype=dtype, device=dev)
        out = {k: (v[:, :n] if layout == "time" else v) for k, v in full.items()}
        out.update(_full=full, _plies=T, _layout=layout, _ply_stride=ply_stride, _tile_stride=tile_stride,
                   _slot_boards=ply_stride if layout == "time" else None, _placement=placed)
        return out

    def _last_ply(self, out: dict, key: str) -> torch.Tensor:
        """The (N, ...) slice of trajectory entry `key` that belongs to the last ply (a view)."""
        T, n = out["_plies"], self.num_envs
        if out["_layout"] == "time":
            return out[key][T - 1]
        v = out[key][:, T - 1]  # (tiles, 64, ...)
        return v.reshape((v.shape[0] * 64,) + tuple(v.shape[2:]))[:n]

    def collect(self, plies: int, out: dict | None = None, count: bool = False, refresh: bool = True,
                layout: str = "time", policies=None, opening_plies: int = 0, first_actions=None, first_status=None) -> dict:
        """``plies`` masked-random plies with auto-reset in ONE launch (``gbl_collect``), EVERY ply materialised:
        entry t of the returned tensors -- "actions", "winner", "rewards", "done", "to_move", "action_mask",
        "observation", each (plies, N, ...) in the default time-major layout -- is what ``rollout(1)`` called ``plies``
        times would have left in the attribute tensors after call t: the action played, its result, and the mask /
        observation of the agent to move next.  (``layout="tile"``, or buffers made with it: (tiles, plies, 64, ...),
        see ``trajectory_buffers``.)  The environment's own tensors (squares, to_move, done, turn, counters) hold the
        position after the last ply; with ``refresh`` the ``action_mask`` / ``observation`` / ``actions`` / ``winner`` /
        ``rewards`` attributes are copied from the last ply (device copies of ~180 B per board: a pure collector that
        only reads the trajectory passes ``refresh=False`` and calls ``refresh()`` before it next steps by hand).
        ``out``: a dict from ``trajectory_buffers(plies)`` to write into (a replay buffer's staging area; placed for speed,
        see there).  Without it the environment's OWN staging buffers are used: one set per (plies, layout, policy outputs),
        made -- and placed, see ``trajectory_buffers`` -- on the first call and reused by every later one, so the returned
        tensors are valid until the next ``collect`` of the same shape on this environment -- two successive calls return THE
        SAME tensors (clone what must outlive the next call, pass ``out``, or pass ``out="fresh"`` for buffers of your own made
        on the spot).  At most ``STAGING_SETS`` (2) sets are kept, least recently used dropped first (a set is several GiB at
        2^20 boards); only ``release_staging()`` drops them all -- ``reset()`` keeps them.  hipGraph caveat: a graph captured around
        ``collect(T)`` WITHOUT ``out=`` has the staging set's addresses baked in; a set that is used while a stream is capturing is
        therefore pinned (exempt from the LRU eviction) and lives until ``release_staging()`` -- call that only once no such graph will
        be replayed again (or capture with ``out=`` buffers of your own, whose lifetime you hold).  Round 3 allocated fresh unplaced
        buffers per call, which ran at 0.72-0.85 of the placed rate at 2^20 boards and paid an allocation per call.

        ``first_actions`` (int (N,)): the first ply plays these actions -- an external policy's decision -- and the
        remaining plies are sampled (``gbl_collect_from``): ``collect(2, out, first_actions=a)`` is one decision of the
        policy plus the masked-random opponent's reply in one launch; the policy reads ``out["observation"][1]`` /
        ``out["action_mask"][1]`` next.  ``first_status`` (int8 (N,), optional): the status byte of those actions, as ``step``'s.

        ``policies=(p1, p2)``: how player_1 / player_2 choose their moves INSIDE the launch (``gbl_collect_policy``) --
        "random" (the masked-uniform sampler, the default for both) or "greedy1" / "greedy" (= "greedy2") / "greedy3": the
        reference's ``GreedyGobbletPolicy.compute_action`` at that depth (greedy_policy.py:38-221) with its fallback draw
        and per-agent action history (``policy_hist``, int8 (N, 2, 3), kept across calls and across games like the
        reference's policy object; ``reset_policy_history()`` clears it).  ``opening_plies``: the first plies of every
        game are drawn at random by a greedy side too (tutorials/GreedyAgent/tutorial_greedy.py:34-41 uses 2; needs
        ``track_turn=True``).  Buffers made with ``policy_outputs=True`` also receive "chosen" / "how" (/ "candidates")."""
        if not self.auto_reset:
            raise ValueError("collect() plays with auto-reset; this environment was created with auto_reset=False")
        T = int(plies)
        if isinstance(out, str):
            if out != "fresh":
                raise ValueError("out: a dict from trajectory_buffers(), None (the environment's staging buffers) or 'fresh'")
            # buffers of the caller's own: made (and placed) now, not kept by the environment, never overwritten by a later call
            out = self.trajectory_buffers(T, layout=layout, policy_outputs=policies is not None, far=False)
        if out is None:
            key = (T, layout, policies is not None)
            out = self._staging.pop(key, None)
            made = out is None
            if made:
                out = self.trajectory_buffers(T, layout=layout, policy_outputs=policies is not None, far=False)
            # (buffers made inside a graph capture belong to the graph's private pool: not kept beyond it)
            capturing = self.device.type == "cuda" and torch.cuda.is_current_stream_capturing()
            if not (made and capturing):
                if capturing:
                    out["_pinned"] = True            # a graph now replays into these addresses: never evicted (release_staging() only)
                self._staging[key] = out             # most recently used last
                loose = [k for k, v in self._staging.items() if not v.get("_pinned")]
                while len(loose) > self.STAGING_SETS:
                    self._staging.pop(loose.pop(0))
        if out["_plies"] != T:
            raise ValueError("trajectory buffers were made for %d plies" % out["_plies"])
        f, n = out["_full"], self.num_envs
        cells = f["actions"].numel()  # (every array of the set has the same number of (ply, board) cells)
        need = (T - 1) * out["_ply_stride"] + (-(-n // 64) - 1) * out["_tile_stride"] + 64
        if f["actions"].device != self.device or cells < need - 64 + (n - 1) % 64 + 1 or ("observation" in f) != (self.observation is not None):
            raise ValueError("trajectory buffers do not fit this environment (made by another one?)")
        if first_actions is not None and policies is not None:
            raise ValueError("first_actions and policies exclude each other")
        fa = None if first_actions is None else _as_i32(first_actions, n, self.device, "first_actions")
        first_status = self._i8_out(first_status, "first_status")
        if first_status is not None and fa is None:
            raise ValueError("first_status needs first_actions")
        if policies is not None:
            try:
                p0, p1 = (self.POLICIES[x] if isinstance(x, str) else int(x) for x in policies)
            except (KeyError, ValueError, TypeError):
                raise ValueError("policies: a pair out of %s" % sorted(self.POLICIES)) from None
            if opening_plies and self.turn is None:
                raise ValueError("opening_plies needs the per-board turn counter: create the environment with track_turn=True")
            if self.policy_hist is None:
                self.reset_policy_history()
            nat.check(self._lib.gbl_collect_policy(
                self.squares.data_ptr(), self.to_move.data_ptr(), self.done.data_ptr(), self.policy_hist.data_ptr(),
                f["actions"].data_ptr(), f["winner"].data_ptr(), f["rewards"].data_ptr(), f["done"].data_ptr(),
                f["to_move"].data_ptr(), f["action_mask"].data_ptr(),
                f["observation"].data_ptr() if "observation" in f else None, nat.ptr(f.get("chosen")), nat.ptr(f.get("how")),
                nat.ptr(f.get("candidates")), n, out["_ply_stride"], out["_tile_stride"], self.seed, self.env_base, self._ply,
                nat.ptr(self._ply_dev), T, p0, p1, int(opening_plies), self.illegal_mode,
                self._counters.data_ptr() if count else None, nat.ptr(self.turn), self._stream()), "gbl_collect_policy")
        else:
            nat.check(self._lib.gbl_collect_from_ex(self.squares.data_ptr(), self.to_move.data_ptr(), self.done.data_ptr(),
                                                    nat.ptr(fa), nat.ptr(first_status), f["actions"].data_ptr(),
                                                    f["winner"].data_ptr(), f["rewards"].data_ptr(), f["done"].data_ptr(),
                                                    f["to_move"].data_ptr(), f["action_mask"].data_ptr(),
                                                    f["observation"].data_ptr() if "observation" in f else None, n,
                                                    out["_ply_stride"], out["_tile_stride"], self.seed, self.env_base,
                                                    self._ply, nat.ptr(self._ply_dev), T, self.illegal_mode,
                                                    self._counters.data_ptr() if count else None, nat.ptr(self.turn),
                                                    self._stream()), "gbl_collect")
        self._ply += T
        if not refresh:
            return out
        # keep the attribute tensors consistent with the position after the last ply
        self.action_mask.copy_(self._last_ply(out, "action_mask")); self.actions.copy_(self._last_ply(out, "actions"))
        self.winner.copy_(self._last_ply(out, "winner")); self.rewards.copy_(self._last_ply(out, "rewards"))
        if self.observation is not None:
            self.observation.copy_(self._last_ply(out, "observation"))
        return out

    def release_staging(self) -> None:
        """Drop the trajectory buffers ``collect()`` keeps for calls without ``out`` (their blocks go back to the driver
        once the last tensor over them is gone; blocks parked during a graph capture are freed here too)."""
        self._staging.clear()
        if self.device.type == "cuda":
            _placement.free_parked()

    def place(self, out: dict, rehome: bool = True, far: bool | None = None) -> dict:
        """Probe -- and, with ``rehome``, re-place -- the observation / mask arrays of trajectory buffers the caller made
        WITHOUT the search (``trajectory_buffers(..., placement="any")``, or buffers whose placement record says the arrays
        share a 96 GiB class of HBM: ratio >= 0.95, 20 % slower at 2^20 boards).  Returns the placement record (also stored
        under ``out["_placement"]``); when a better pair is found the dict's two arrays are REPLACED by it (contents copied,
        views rebuilt) -- tensors taken from the dict before the call keep pointing at the old arrays."""
        f = out["_full"]
        if "observation" not in f or f["action_mask"].numel() < _placement.MIN_BYTES:
            rec = {"spread": False, "why": "arrays too small to probe" if "observation" in f else "no observation stream"}
            out["_placement"] = rec
            return rec
        if self.device.type == "cuda" and torch.cuda.is_current_stream_capturing():
            raise ValueError("place() probes and synchronises: not inside a graph capture")
        geometry = (dict(slot_boards=out["_ply_stride"], plies=out["_plies"])
                    if out["_layout"] == "time" and out["_ply_stride"] % 128 == 0 else {})
        obs, mask = f["observation"], f["action_mask"]
        keep_o, keep_m = obs.clone(), mask.clone()  # (a probe overwrites what it measures)
        flat = lambda t: t.view(torch.uint8).reshape(-1)  # noqa: E731
        if not rehome:
            both, ua, ub = _placement.probe(flat(obs), flat(mask), **geometry)
            rec = {"ratio": round(both / max(ua + ub, 1e-9), 3), "probes": [round(both / max(ua + ub, 1e-9), 3)],
                   "ended": "probed only"}
            rec["spread"] = rec["ratio"] <= _placement.SPREAD_RATIO
            obs.copy_(keep_o); mask.copy_(keep_m)
        else:
            a, b, rec = _placement.spread_pair(obs.numel(), mask.numel(), self.device, **geometry, far=far,
                                               plain=lambda: (flat(obs), flat(mask)))
            f["observation"] = a.view(torch.int8).view(obs.shape)
            f["action_mask"] = b.view(torch.int8).view(mask.shape)
            f["observation"].copy_(keep_o); f["action_mask"].copy_(keep_m)
            n = self.num_envs
            for k in ("observation", "action_mask"):
                out[k] = f[k][:, :n] if out["_layout"] == "time" else f[k]
        out["_placement"] = rec
        return rec

    def step_into(self, actions, out: dict, t: int, status=None, next_actions=None):
        """``step(actions)`` with this ply's outputs written straight into slot ``t`` of time-major trajectory buffers
        (``trajectory_buffers``) -- the collector loop of a policy that lives outside the library (the reference's
        Tianshou / RLlib training loops: policy(obs, mask) -> env.step -> buffer.add), without a copy per ply:
        ``out["action_mask"][t]``, ``["observation"][t]``, ``["winner"][t]``, ``["rewards"][t]`` come from the kernel,
        and so do ``["actions"][t]``, ``["done"][t]``, ``["to_move"][t]`` (``gbl_step_into``: one launch).  Returns the views
        (observation[t], action_mask[t]) the policy reads for the next ply.  The environment's own ``action_mask`` /
        ``observation`` / ``winner`` / ``rewards`` attributes are NOT updated (call ``refresh()`` before stepping by
        hand again); ``squares`` / ``to_move`` / ``done`` / ``turn`` are, as always.  ``status`` / ``next_actions``: as ``step``'s
        (``gbl_step_ex``)."""
        if out["_layout"] != "time":
            raise ValueError("step_into() writes time-major trajectory buffers")
        t = int(t)
        if not 0 <= t < out["_plies"]:
            raise IndexError("slot %d of a %d-ply trajectory" % (t, out["_plies"]))
        f, n = out["_full"], self.num_envs
        if f["actions"].device != self.device or f["actions"].shape[1] < n or ("observation" in f) != (self.observation is not None):
            raise ValueError("trajectory buffers do not fit this environment (made by another one?)")
        a = _as_i32(actions, n, self.device, "actions")
        obs_t = f["observation"][t] if "observation" in f else None
        status, next_actions = self._i8_out(status, "status"), self._i32_out(next_actions, "next_actions")
        nat.check(self._lib.gbl_step_ex(self.squares.data_ptr(), self.to_move.data_ptr(), self.done.data_ptr(),
                                        a.data_ptr(), f["winner"][t].data_ptr(), f["rewards"][t].data_ptr(),
                                        f["action_mask"][t].data_ptr(), nat.ptr(obs_t), nat.ptr(self.turn),
                                        f["actions"][t].data_ptr(), f["done"][t].data_ptr(), f["to_move"][t].data_ptr(),
                                        nat.ptr(status), nat.ptr(next_actions), self.seed, self.env_base, self._ply + 1,
                                        nat.ptr(self._ply_dev), n, self.illegal_mode, int(self.auto_reset), self._stream()),
                  "gbl_step_into")
        self._ply += 1
        return (out["observation"][t] if obs_t is not None else None), out["action_mask"][t]

    # -- masked-uniform sampling (examples/example_basic.py:58-61) ----------------------------------------
    def sample_actions(self, out: torch.Tensor | None = None) -> torch.Tensor:
        """One uniformly random legal action per board from the current action_mask, keyed by
        (seed, env_base + b, ply)."""
        out = self.actions if out is None else out
        nat.check(self._lib.gbl_sample_at(self.action_mask.data_ptr(), out.data_ptr(), self.num_envs, self.seed,
                                          self.env_base, self._ply, nat.ptr(self._ply_dev), self._stream()),
                  "gbl_sample")
        return out

    def rollout(self, plies: int = 1, count: bool = False):
        """``plies`` masked-random plies with auto-reset in ONE kernel launch (sample + step fused;
        the boards stay in registers between plies).  The attribute tensors hold the outputs of
        the last ply.  ``count=True`` also accumulates ``counters`` (device atomics, a few
        microseconds per launch).  ``rollout(1)`` is one ply of the benchmark pipeline with every
        output materialised."""
        if not self.auto_reset:
            raise ValueError("rollout() plays with auto-reset (finished boards start a new game); this environment was "
                             "created with auto_reset=False -- use sample_actions() + step() to keep finished boards frozen")
        n = self.num_envs
        nat.check(self._lib.gbl_rollout_at(self.squares.data_ptr(), self.to_move.data_ptr(), self.done.data_ptr(),
                                           self.actions.data_ptr(), self.winner.data_ptr(), self.rewards.data_ptr(),
                                           self.action_mask.data_ptr(), nat.ptr(self.observation), n, self.seed,
                                           self.env_base, self._ply, nat.ptr(self._ply_dev), int(plies),
                                           self.illegal_mode, self._counters.data_ptr() if count else None,
                                           nat.ptr(self.turn), self._stream()),
                  "gbl_rollout")
        self._ply += int(plies)
        return self.observe(), self.rewards, self.done, self.winner
