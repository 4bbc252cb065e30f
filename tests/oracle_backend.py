"""TEST-ONLY stand-in for the facade's private 1-board engine (``gobblet_v1._HipBoardEngine``) backed by the CPU
oracle, so that the host logic of ``gobblet_v1.Board`` / ``raw_env`` / ``env()`` can be exercised without a GPU.
The product never imports this; its engine is always the HIP library."""
import numpy as np

import oracle


class OracleBoardEngine:
    def evaluate(self, squares, agent_index=None, action=None) -> dict:
        sq = np.ascontiguousarray(squares, dtype=np.int8).reshape(1, 27).copy()
        if action is not None and 0 <= action < 54:
            sq[0] = oracle.play_turn(sq[0], agent_index, action)
        who = [np.zeros(1, np.int8), np.ones(1, np.int8)]
        return {"squares": sq[0].copy(), "winner": oracle.batch_winner(sq).astype(np.int8),
                "flat": oracle.batch_flatboard(sq)[0], "covered": oracle.batch_covered(sq)[0],
                "mask0": oracle.batch_legal_mask(sq, who[0])[0], "mask1": oracle.batch_legal_mask(sq, who[1])[0],
                "obs0": oracle.batch_observe(sq, who[0], 0)[0].reshape(-1),
                "obs1": oracle.batch_observe(sq, who[1], 1)[0].reshape(-1)}
