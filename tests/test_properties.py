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
    # planes_to_row(make_planes(row)) is the identity on contract states: k_collect5 keeps its boards as bit planes for a whole
    # launch and rebuilds the 27-byte rows from them once, at the end
    assert np.array_equal(emu.planes_roundtrip(sq), sq)
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


@settings(max_examples=25, **RELAXED)
@given(valid_boards(), st.sampled_from(["noop", "terminate"]), st.booleans(), st.integers(0, 2 ** 40), st.integers(0, 1000))
def test_step_ex_status_and_fused_draw(data, illegal_mode, auto_reset, seed, ply):
    """gbl_step_ex (host flavour: the device header's own lane functions) on ARBITRARY contract boards: the status byte is 0 exactly
    for actions inside the mover's legal mask, carries bit 1 exactly for indices outside [0, 54), is 0 on frozen boards whatever the
    action; the fused draw is gbl_sample's rule on the mask the step stores (-1 where nobody is to move)."""
    import torch

    import gobblet_rl_amd as G
    sq, tm, rng = data
    n = len(sq)
    dn = (oracle.batch_winner(sq) != 0).astype(np.int8) if not auto_reset else np.zeros(n, np.int8)
    a = rng.integers(-3, 58, n).astype(np.int32)
    legal_before = oracle.batch_legal_mask(sq, tm)
    env = G.BatchedGobblet(n, "cpu", illegal_mode=illegal_mode, auto_reset=auto_reset, seed=seed)
    env.board.squares = torch.from_numpy(sq.copy()); env.to_move.copy_(torch.from_numpy(tm)); env.done.copy_(torch.from_numpy(dn))
    env.refresh()
    env.ply = ply
    status = torch.full((n,), 9, dtype=torch.int8)
    nxt = torch.full((n,), 9, dtype=torch.int32)
    acts = torch.from_numpy(a.copy())
    exp_status = oracle.batch_action_status(sq, tm, dn, a, auto_reset=auto_reset)
    obs, rew, done, win = env.step(acts, status=status, next_actions=nxt)
    st8 = status.numpy()
    assert np.array_equal(st8, exp_status)
    frozen = (dn != 0) & (not auto_reset)
    in_range = (a >= 0) & (a < 54)
    legal = in_range & (legal_before[np.arange(n), np.clip(a, 0, 53)] != 0)
    assert np.array_equal(st8 == 0, legal | frozen) and np.array_equal((st8 & 2) != 0, ~in_range & ~frozen)
    s1, t1, d1 = sq.copy(), tm.copy(), dn.copy()
    o = oracle.batch_step(s1, t1, d1, a, illegal_mode=0 if illegal_mode == "noop" else 1, auto_reset=auto_reset)
    assert np.array_equal(env.squares.numpy(), s1) and np.array_equal(obs["action_mask"].numpy(), o["mask"])
    exp_next = oracle.batch_sample(o["mask"], seed, 0, ply + 1)
    assert np.array_equal(nxt.numpy(), exp_next) and ((exp_next == -1) == (o["mask"].sum(1) == 0)).all()
    # the draw over the action array itself: same result
    env2 = G.BatchedGobblet(n, "cpu", illegal_mode=illegal_mode, auto_reset=auto_reset, seed=seed)
    env2.board.squares = torch.from_numpy(sq.copy()); env2.to_move.copy_(torch.from_numpy(tm)); env2.done.copy_(torch.from_numpy(dn))
    env2.refresh()
    env2.ply = ply
    env2.step(acts, next_actions=acts)
    assert np.array_equal(acts.numpy(), exp_next)
