"""``GreedyGobbletPolicy`` -- the reference's depth-1/2/3 lookahead policy (gobblet_rl/game/greedy_policy.py)
for N boards at once on the GPU.

Same constructor and method names as the reference class.  ``compute_action(obs, mask)`` takes one
observation like the reference (and returns a numpy scalar); ``compute_actions(obs, mask)`` takes
batches (torch tensors on the device, or numpy) and returns an int32 tensor (N,).  The board is
rebuilt from the observation exactly as greedy_policy.py:43-71 does (``gbl_decode_obs``); the search
is ``gbl_greedy``.  Per-agent history of own actions (``prev_actions``, greedy_policy.py:19,211-219) is
kept per board on the device.  Where the reference falls back to ``np.random.choice(actions_depth1)``
(:211-217, numpy's global RNG) this class draws uniformly from the same candidate set with the
library's counter-based sampler (the ``gbl_sample`` rule on a generator stream of its own), keyed by
(seed, global board id, call index).
``depth=3`` is accepted and decides like ``depth=2``: the reference's depth-3 block (:160-208) only ever
re-assigns ``chosen_action = action``, which :157 has just done (checked against the reference itself,
tests/golden/greedy_depth3.npz).
"""
from __future__ import annotations

from typing import Any, Optional

import numpy as np
import torch

from . import _native as nat


class GreedyGobbletPolicy:
    def __init__(self, depth: Optional[int] = 2, seed: Optional[int] = 0, device="cuda:0", env_base: int = 0,
                 **kwargs: Any) -> None:
        """env_base: global index of board 0 of the batches this policy is handed (a shard of a larger batch passes
        its first global board, like ``BatchedGobblet``): the fallback draw is keyed by the global board id, so
        trajectories do not depend on how the boards are sharded."""
        if depth not in (1, 2, 3):
            raise ValueError("depth must be 1, 2 or 3")
        self.depth = depth
        self.seed = int(seed or 0)
        self.env_base = int(env_base)
        self.device = torch.device(device)
        self._lib = nat.lib_for(self.device)  # ("cpu": the host flavour of the ABI, asked for -- never a fallback)
        self.prev_actions = None  # int8 (N, 2, 3) on device: last three own actions per agent, -1 = none
        self._calls, self._calls_dev = 0, None  # call index (keys the fallback draw); see device_calls()
        # outputs of the last call (device tensors): chosen-or--1, candidate set, fallback flag
        self.last_chosen = self.last_candidates = self.last_fallback = None

    def _stream(self):
        return nat.current_stream(self.device)

    def _ensure(self, n):
        if self.prev_actions is None or self.prev_actions.shape[0] != n:
            self.prev_actions = torch.full((n, 2, 3), -1, dtype=torch.int8, device=self.device)

    def reset_history(self):
        self.prev_actions = None

    def device_calls(self, enable: bool = True) -> None:
        """Keep the call index in device memory (like ``BatchedGobblet.device_ply``): a captured hipGraph that
        ends with ``advance_calls()`` then draws fresh fallback actions on every replay."""
        if enable and self._calls_dev is None:
            self._calls_dev = torch.full((1,), self._calls, dtype=torch.int32, device=self.device)
            self._calls = 0
        elif not enable and self._calls_dev is not None:
            self._calls, self._calls_dev = self._calls + int(self._calls_dev.item()), None

    def advance_calls(self) -> None:
        if self._calls_dev is not None and self._calls:
            nat.check(self._lib.gbl_counter_add(self._calls_dev.data_ptr(), self._calls, self._stream()),
                      "gbl_counter_add")
            self._calls = 0

    def compute_actions(self, obs, mask=None) -> torch.Tensor:
        """obs: int8 (N,3,3,13); mask: int8 (N,54) or None (derive the legal mask from the board)."""
        obs = torch.as_tensor(obs).to(device=self.device, dtype=torch.int8).reshape(-1, 3, 3, 13).contiguous()
        n = obs.shape[0]
        if mask is not None:
            mask = torch.as_tensor(mask).to(device=self.device, dtype=torch.int8).reshape(n, nat.ACTIONS).contiguous()
        state = torch.empty((n, nat.CELLS), dtype=torch.int8, device=self.device)
        who = torch.empty(n, dtype=torch.int8, device=self.device)
        nat.check(self._lib.gbl_decode_obs(obs.data_ptr(), state.data_ptr(), who.data_ptr(), n, self._stream()),
                  "gbl_decode_obs")  # greedy_policy.py:43-71
        return self.compute_actions_from_state(state, who, mask)

    def compute_actions_from_state(self, state: torch.Tensor, to_move: torch.Tensor, mask=None) -> torch.Tensor:
        """The same decision from ``squares`` (N,27) + ``to_move`` (N,) directly (no observation round trip)."""
        n = state.shape[0]
        self._ensure(n)
        out = torch.empty(n, dtype=torch.int32, device=self.device)
        act = torch.empty(n, dtype=torch.int32, device=self.device)
        cand = torch.empty((n, nat.ACTIONS), dtype=torch.int8, device=self.device)
        fb = torch.empty(n, dtype=torch.int8, device=self.device)
        # one launch: the search, the :211-217 fallback draw (uniform over actions_depth1, keyed by seed / board /
        # call index) and the :219 history append for the acting agent
        nat.check(self._lib.gbl_greedy_act_at(state.data_ptr(), to_move.data_ptr(), nat.ptr(mask),
                                              self.prev_actions.data_ptr(), self.depth, self.seed, self.env_base,
                                              self._calls,
                                              nat.ptr(self._calls_dev), out.data_ptr(), act.data_ptr(),
                                              cand.data_ptr(), fb.data_ptr(), n, self._stream()), "gbl_greedy_act")
        self.last_chosen, self.last_candidates, self.last_fallback = act, cand, fb
        self._calls += 1
        return out

    # -- reference-shaped single-observation entry points ------------------------------------------------
    def compute_action(self, obs, mask) -> np.ndarray:  # greedy_policy.py:38-221
        return np.array(int(self.compute_actions(np.asarray(obs)[None], np.asarray(mask)[None])[0]))

    def compute_actions_rllib(self, obs_batch):  # greedy_policy.py:21-31
        observations = np.asarray(obs_batch["observation"])
        observations = observations.reshape(observations.shape[0], 3, 3, -1)
        return list(self.compute_actions(observations, np.asarray(obs_batch["action_mask"])).cpu().numpy())

    def forward(self, batch, state=None, **kwargs):
        """Tianshou-adapter shape (greedy_policy_tianshou.py:63-84): ``batch.obs.obs`` / ``batch.obs.mask``
        (or dict keys "obs" / "mask") for all environments at once -> {"act": int64 (N,)} on the host."""
        ob = batch["obs"] if isinstance(batch, dict) else batch.obs
        obs = ob["obs"] if isinstance(ob, dict) else ob.obs
        mask = ob["mask"] if isinstance(ob, dict) else ob.mask
        act = self.compute_actions(obs, torch.as_tensor(mask).to(torch.int8))
        return {"act": act.to(torch.int64).cpu().numpy()}

    def compute_action_tianshou(self, obs):  # greedy_policy.py:33-36
        mask = obs.mask
        obs = obs.obs if hasattr(obs, "obs") else obs
        return self.compute_action(obs, mask)
