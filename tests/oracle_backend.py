"""TEST-ONLY stand-in for ``BatchedBoard`` backed by the CPU oracle, so that the host logic of
``gobblet_v1.Board`` / ``raw_env`` / ``env()`` can be exercised without a GPU.  The product never
imports this; its default backend is the HIP engine."""
import numpy as np
import torch

import oracle


class OracleBoardBackend:
    def __init__(self, num_envs=1):
        self.num_envs = num_envs
        self._sq = np.zeros((num_envs, 27), np.int8)
        self.calculate_winners()

    def calculate_winners(self):
        self.winning_combinations = [(0, 1, 2), (3, 4, 5), (6, 7, 8), (0, 3, 6), (1, 4, 7), (2, 5, 8), (0, 4, 8),
                                     (2, 4, 6)]

    def setup(self):
        self.calculate_winners()

    @property
    def squares(self):
        return torch.from_numpy(self._sq)

    @squares.setter
    def squares(self, v):
        self._sq = np.ascontiguousarray(torch.as_tensor(v).numpy().astype(np.int8).reshape(self.num_envs, 27))

    def _b(self, x):
        return np.broadcast_to(np.asarray(x), (self.num_envs,))

    def is_legal(self, action, agent_index=0):
        a, g = self._b(action), self._b(agent_index)
        return torch.tensor([0 <= a[i] < 54 and oracle.is_legal(self._sq[i], a[i], g[i]) for i in range(self.num_envs)])

    def legal_mask(self, agent_index):
        return torch.from_numpy(oracle.batch_legal_mask(self._sq, self._b(agent_index).astype(np.int8).copy()))

    def play_turn(self, agent_index, action):
        a, g = self._b(action), self._b(agent_index)
        for i in range(self.num_envs):
            if 0 <= a[i] < 54:
                self._sq[i] = oracle.play_turn(self._sq[i], g[i], a[i])

    def get_action(self, pos, piece_size, agent_index):
        p, s, g = self._b(pos), self._b(piece_size), self._b(agent_index)
        return torch.tensor([oracle.get_action(self._sq[i], p[i], s[i], g[i]) for i in range(self.num_envs)])

    def get_flatboard(self):
        return torch.from_numpy(oracle.batch_flatboard(self._sq))

    def check_for_winner(self):
        return torch.from_numpy(oracle.batch_winner(self._sq))

    def check_covered(self):
        return torch.from_numpy(oracle.batch_covered(self._sq))

    def observation(self, agent_index):
        if isinstance(agent_index, int):
            return torch.from_numpy(oracle.batch_observe(self._sq, np.zeros(self.num_envs, np.int8), agent_index))
        return torch.from_numpy(oracle.batch_observe(self._sq, self._b(agent_index).astype(np.int8).copy(), -1))
