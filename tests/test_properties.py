"""Property tests (CPU): the device code (host emulation) against the oracle on states drawn by
hypothesis -- arbitrary boards obeying the state contract (every piece at most once, on its own
level), not only positions self-play reaches -- plus algebraic properties the game rules imply."""
import numpy as np
from hypothesis import HealthCheck, given, settings, strategies as st

RELAXED = dict(deadline=None, suppress_health_check=list(HealthCheck), derandomize=True)

import oracle
from tests import emu


@st.composite
def valid_boards(draw, n_min=1, n_max=130):
    n = draw(st.integers(n_min, n_max))
    sq = np.zeros((n, 27), np.int8)
    seed = draw(st.integers(0, 2 ** 31 - 1))
    rng = np.random.default_rng(seed)
    dens = draw(st.floats(0.0, 1.0))
    for b in range(n):
        for piece in range(1, 7):
            for sign in (1, -1):
                if rng.random() < dens:
                    level = (piece - 1) // 2
                    free = np.flatnonzero(sq[b, 9 * level:9 * level + 9] == 0)
                    sq[b, 9 * level + rng.choice(free)] = sign * piece
    tm = rng.integers(0, 2, n).astype(np.int8)
    return sq, tm, rng


@settings(max_examples=40, **RELAXED)
@given(valid_boards())
def test_board_functions_match_oracle(data):
    sq, tm, rng = data
    assert np.array_equal(emu.flatboard(sq), oracle.batch_flatboard(sq))
    assert np.array_equal(emu.covered(sq), oracle.batch_covered(sq))
    assert np.array_equal(emu.winner(sq), oracle.batch_winner(sq))
    m = emu.legal_mask(sq, tm)
    assert np.array_equal(m, oracle.batch_legal_mask(sq, tm))
    assert np.array_equal(emu.observe(sq, tm, -1), oracle.batch_observe(sq, tm, -1))
    # rules: a large piece is never covered, so an agent holding/showing one always has a move
    assert (m[:, 36:].sum(1) > 0).all() or (np.abs(sq[:, 18:]) > 0).sum(1).max() == 9
    # decode(observe(board)) is the identity (greedy_policy.py:43-71 inverts gobblet.py:179-208)
    st_, who = emu.decode_obs(oracle.batch_observe(sq, tm, -1))
    assert np.array_equal(st_, sq) and np.array_equal(who, tm)


@settings(max_examples=25, **RELAXED)
@given(valid_boards(), st.sampled_from([0, 1]), st.booleans())
def test_step_matches_oracle_and_preserves_invariants(data, illegal_mode, auto_reset):
    sq, tm, rng = data
    n = len(sq)
    dn = (oracle.batch_winner(sq) != 0).astype(np.int8) if not auto_reset else np.zeros(n, np.int8)
    a = rng.integers(-1, 55, n).astype(np.int32)
    s1, t1, d1 = sq.copy(), tm.copy(), dn.copy()
    s2, t2, d2 = sq.copy(), tm.copy(), dn.copy()
    o1 = oracle.batch_step(s1, t1, d1, a, illegal_mode=illegal_mode, auto_reset=auto_reset)
    o2 = emu.step(s2, t2, d2, a, illegal_mode=illegal_mode, auto_reset=auto_reset)
    assert np.array_equal(s1, s2) and np.array_equal(t1, t2) and np.array_equal(d1, d2)
    for k in o1:
        assert np.array_equal(o1[k], o2[k]), k
    # a step moves at most one piece: the multiset of pieces only grows by the placed one
    for b in range(n):
        before, after = sq[b][sq[b] != 0], s2[b][s2[b] != 0]
        if auto_reset and d2[b]:
            assert after.size == 0
        else:
            assert len(after) - len(before) in (0, 1) and len(np.unique(after)) == len(after)


@settings(max_examples=12, **RELAXED)
@given(valid_boards(n_max=70))
def test_greedy_matches_oracle(data):
    sq, tm, rng = data
    live = oracle.batch_winner(sq) == 0
    if not live.any():
        return
    sq, tm = np.ascontiguousarray(sq[live]), np.ascontiguousarray(tm[live])
    for depth in (1, 2):
        e = emu.greedy(sq, tm, depth=depth)
        o = oracle.batch_greedy(sq, tm, depth=depth)
        for x, y in zip(e, o):
            assert np.array_equal(x, y)
