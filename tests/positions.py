"""Position generators shared by the CPU and the GPU tests (test infrastructure; uses the oracle only)."""
import numpy as np

import oracle


def terminal_roots(boards, n_random=20000, seed=4):
    """Positions on which somebody holds a line already (the reference's policy is never asked about them in a game, but
    the entry points take any board): the golden set's, plus random stacks."""
    rng = np.random.default_rng(seed)
    st = np.zeros((n_random, 27), np.int8)
    for lvl, vals in enumerate(((1, 2), (3, 4), (5, 6))):  # one piece of each number per colour at most: a valid board
        for v in vals:
            for sign in (1, -1):
                put = rng.random(n_random) < 0.55
                pos = rng.integers(0, 9, n_random)
                free = st[np.arange(n_random), 9 * lvl + pos] == 0
                ok = put & free
                st[np.flatnonzero(ok), 9 * lvl + pos[ok]] = sign * v
    st = st[oracle.batch_winner(st) != 0]
    return np.ascontiguousarray(np.concatenate([boards["squares"][boards["winner"] != 0], st]))
