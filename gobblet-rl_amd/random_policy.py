"""``RandomAdmissiblePolicy`` -- uniform over the legal actions, batched on the device: the rule of the
reference's ``random_admissible_policy_rllib.py:23-30`` (``np.random.choice(54, p=mask/sum(mask))``) and of
``examples/example_basic.py:58-61``, drawn with the library's counter-based sampler (``gbl_sample``) so that
it is reproducible and identical on CPU oracle and GPU."""
from __future__ import annotations

import torch

from . import _native as nat


class RandomAdmissiblePolicy:
    def __init__(self, seed: int = 0, device="cuda:0", env_base: int = 0):
        self.seed, self.env_base = int(seed), int(env_base)
        self.device = torch.device(device)
        self._lib = nat.lib_for(self.device)  # ("cpu": the host flavour of the ABI, asked for -- never a fallback)
        self._calls, self._calls_dev = 0, None

    def device_calls(self, enable: bool = True) -> None:
        """Call index in device memory, for hipGraph replay (see ``BatchedGobblet.device_ply``)."""
        if enable and self._calls_dev is None:
            self._calls_dev = torch.full((1,), self._calls, dtype=torch.int32, device=self.device)
            self._calls = 0
        elif not enable and self._calls_dev is not None:
            self._calls, self._calls_dev = self._calls + int(self._calls_dev.item()), None

    def advance_calls(self) -> None:
        if self._calls_dev is not None and self._calls:
            nat.check(self._lib.gbl_counter_add(self._calls_dev.data_ptr(), self._calls,
                                                nat.current_stream(self.device)), "gbl_counter_add")
            self._calls = 0

    def compute_actions(self, obs_batch, **kwargs) -> torch.Tensor:
        """obs_batch: {"action_mask": (N,54) int8, ...} (RLlib-style) or the mask tensor itself -> int32 (N,)."""
        mask = obs_batch["action_mask"] if isinstance(obs_batch, dict) else obs_batch
        mask = torch.as_tensor(mask).to(device=self.device, dtype=torch.int8).reshape(-1, nat.ACTIONS).contiguous()
        n = mask.shape[0]
        out = torch.empty(n, dtype=torch.int32, device=self.device)
        nat.check(self._lib.gbl_sample_at(mask.data_ptr(), out.data_ptr(), n, self.seed, self.env_base, self._calls,
                                          nat.ptr(self._calls_dev), nat.current_stream(self.device)), "gbl_sample")
        self._calls += 1
        return out
