"""CPU check of the DEVICE code: gobblet-rl_amd/csrc/gobblet_device.h compiled for the host
(tests/emu) against the oracle, on the golden vectors and on seeded self-play states.
The GPU parity tests (tests/test_gpu_parity.py, -m gpu) run the same comparisons through the
real kernels and the C-ABI."""
import os

import numpy as np
import pytest

import oracle
from tests import emu
from tests.positions import terminal_roots


def selfplay_states(n, plies, seed, p_illegal=0.05):
    """n boards advanced by the oracle's fused rollout to varied game phases."""
    rng = np.random.default_rng(seed)
    state, tm, dn = oracle.batch_reset(n)
    # stagger: board b plays (b % plies) plies so that all phases are present
    for t in range(plies):
        m = oracle.batch_legal_mask(state, tm)
        a = oracle.batch_sample(m, seed, 0, t)
        live = (np.arange(n) % plies) > t
        a = np.where(live, a, -1).astype(np.int32)  # -1: out of range -> illegal no-op... keep those boards as is
        s2, t2, d2 = state.copy(), tm.copy(), dn.copy()
        oracle.batch_step(s2, t2, d2, a)
        state[live], tm[live], dn[live] = s2[live], t2[live], d2[live]
    return state, tm, dn, rng


@pytest.fixture(scope="module")
def boards(golden_dir):
    return np.load(os.path.join(golden_dir, "board_functions.npz"))


@pytest.fixture(scope="module")
def games(golden_dir):
    return np.load(os.path.join(golden_dir, "random_games.npz"))


@pytest.mark.parametrize("n", [1, 3, 63, 64, 65, 408])
def test_board_functions_vs_golden(boards, n):
    sq = np.ascontiguousarray(boards["squares"][:n])
    assert np.array_equal(emu.flatboard(sq), boards["flatboard"][:n])
    assert np.array_equal(emu.covered(sq), boards["covered"][:n])
    assert np.array_equal(emu.winner(sq), boards["winner"][:n])
    z, o = np.zeros(n, np.int8), np.ones(n, np.int8)
    assert np.array_equal(emu.legal_mask(sq, z), boards["legal_p1"][:n])
    assert np.array_equal(emu.legal_mask(sq, o), boards["legal_p2"][:n])
    assert np.array_equal(emu.observe(sq, z, 0), boards["obs_p1"][:n])
    assert np.array_equal(emu.observe(sq, z, 1), boards["obs_p2"][:n])
    assert np.array_equal(emu.observe(sq, o, -1), boards["obs_p2"][:n])


def test_step_vs_golden_games(games):
    g = games
    state = g["squares_before"].copy(); tm = g["mover"].copy(); dn = np.zeros(len(tm), np.int8)
    out = emu.step(state, tm, dn, g["action"])
    assert np.array_equal(state, g["squares_after"]) and np.array_equal(tm, g["to_move_after"])
    assert np.array_equal(out["winner"], g["winner"]) and np.array_equal(dn, g["done"])
    assert np.array_equal(out["reward"], g["reward"])
    live = g["done"] == 0
    assert np.array_equal(out["mask"][live], g["mask_next"][live]) and (out["mask"][~live] == 0).all()
    obs_next = np.where(g["to_move_after"][:, None, None, None] == 0, g["obs_p1"], g["obs_p2"])
    assert np.array_equal(out["obs"], obs_next)


@pytest.mark.parametrize("n,illegal_mode,auto_reset", [(4096, 0, False), (1000, 0, True), (777, 1, False), (130, 1, True)])
def test_step_vs_oracle_selfplay(n, illegal_mode, auto_reset):
    state, tm, dn, rng = selfplay_states(n, 24, seed=n)
    for t in range(6):
        m = oracle.batch_legal_mask(state, tm)
        a = oracle.batch_sample(m, 5, 0, 100 + t)
        wild = rng.random(n) < 0.15
        a = np.where(wild, rng.integers(-3, 58, n), a).astype(np.int32)
        s1, t1, d1 = state.copy(), tm.copy(), dn.copy()
        s2, t2, d2 = state.copy(), tm.copy(), dn.copy()
        o1 = oracle.batch_step(s1, t1, d1, a, illegal_mode=illegal_mode, auto_reset=auto_reset)
        o2 = emu.step(s2, t2, d2, a, illegal_mode=illegal_mode, auto_reset=auto_reset)
        assert np.array_equal(s1, s2) and np.array_equal(t1, t2) and np.array_equal(d1, d2)
        for k in o1:
            assert np.array_equal(o1[k], o2[k]), k
        state, tm, dn = s1, t1, d1
    assert dn.sum() > 0 or auto_reset


def test_board_api_vs_oracle_selfplay():
    n = 2000
    state, tm, dn, rng = selfplay_states(n, 30, seed=11)
    assert np.array_equal(emu.legal_mask(state, tm), oracle.batch_legal_mask(state, tm))
    assert np.array_equal(emu.legal_mask(state, 1 - tm), oracle.batch_legal_mask(state, (1 - tm).astype(np.int8)))
    assert np.array_equal(emu.winner(state), oracle.batch_winner(state))
    assert np.array_equal(emu.flatboard(state), oracle.batch_flatboard(state))
    assert np.array_equal(emu.covered(state), oracle.batch_covered(state))
    assert np.array_equal(emu.observe(state, tm, -1), oracle.batch_observe(state, tm, -1))
    a = rng.integers(-2, 56, n).astype(np.int32)
    ag = rng.integers(0, 2, n).astype(np.int8)
    exp = np.array([(0 <= a[i] < 54) and oracle.is_legal(state[i], a[i], ag[i]) for i in range(n)], np.int8)
    assert np.array_equal(emu.is_legal(state, ag, a), exp)
    s2 = state.copy()
    emu.play_turn(s2, ag, a)
    exp_s = np.stack([oracle.play_turn(state[i], ag[i], a[i]) if 0 <= a[i] < 54 else state[i] for i in range(n)])
    assert np.array_equal(s2, exp_s)


def test_sampler_and_decode():
    n = 1500
    state, tm, dn, rng = selfplay_states(n, 20, seed=4)
    m = oracle.batch_legal_mask(state, tm)
    m[::97] = 0  # empty masks -> -1
    assert np.array_equal(emu.sample(m, 9, 12345678901234, 7), oracle.batch_sample(m, 9, 12345678901234, 7))
    for base in (0, (1 << 32) - 30, (5 << 32) - 700):  # board ids below, across and above 2^32 (derived key)
        assert np.array_equal(emu.sample(m, 2**63 + 11, base, 4000000000), oracle.batch_sample(m, 2**63 + 11, base, 4000000000))
    obs = oracle.batch_observe(state, tm, -1)
    st, who = emu.decode_obs(obs)
    assert np.array_equal(st, state) and np.array_equal(who, tm)


@pytest.mark.parametrize("n,plies", [(700, 1), (1000, 37)])
def test_rollout_vs_oracle(n, plies):
    s1, t1, d1 = oracle.batch_reset(n)
    s2, t2, d2 = oracle.batch_reset(n)
    o1 = oracle.batch_rollout(s1, t1, d1, 42, 5000, 3, plies)
    o2 = emu.rollout(s2, t2, d2, 42, 5000, 3, plies)
    assert np.array_equal(s1, s2) and np.array_equal(t1, t2) and np.array_equal(d1, d2)
    for k in o1:
        assert np.array_equal(o1[k], o2[k]), k


@pytest.mark.parametrize("lpb", [4, 2, 1])
@pytest.mark.parametrize("n,plies,illegal", [(1, 9, 0), (15, 20, 1), (16, 1, 0), (17, 33, 0), (31, 5, 0), (33, 7, 1), (63, 12, 1),
                                             (64, 3, 0), (65, 40, 0), (1000, 25, 0)])
def test_small_batch_walk_vs_oracle(n, plies, illegal, lpb):
    """The role kernel's device code on the host (k_collect_small<.., LPB>: sub-tiles of 64 / LPB boards, LPB lanes per board that
    play alike and share a mask row -- bytes [64 j / LPB, 64 (j + 1) / LPB) -- and an observation row -- channels j, j + LPB, ...):
    sub-tile load / store incl. ragged last sub-tiles, the images' way out through sub_fetch / sub_store, the row builders of
    every LPB, the byte patches of the state image; against the oracle."""
    s1, t1, d1 = oracle.batch_reset(n)
    s2, t2, d2 = oracle.batch_reset(n)
    o1 = oracle.batch_rollout(s1, t1, d1, 7, 123456789012, 5, plies, illegal_mode=illegal)
    o2 = emu.rollout_small(s2, t2, d2, 7, 123456789012, 5, plies, illegal_mode=illegal, lpb=lpb)
    assert np.array_equal(s1, s2) and np.array_equal(t1, t2) and np.array_equal(d1, d2)
    for k in o2:
        assert np.array_equal(o1[k], o2[k]), k


def test_risky_squares_as_threat_squares():
    """greedy_risky_squares' core -- the squares that complete a line inside a 9-bit set, or all nine if it holds one -- as threat
    squares (risky_from_have) against the walk over the eight lines, on all 512 sets."""
    assert emu.lib().emu_risky_mismatches() == 0


@pytest.mark.parametrize("depth", [1, 2])
def test_greedy_vs_golden_and_oracle(golden_dir, depth):
    g = np.load(os.path.join(golden_dir, "greedy.npz"))
    sq = np.ascontiguousarray(g["squares"]); tm = np.ascontiguousarray(g["to_move"])
    act, cm, fb = emu.greedy(sq, tm, depth=depth)
    assert np.array_equal(act, g[f"chosen_d{depth}"].astype(np.int32))
    assert np.array_equal(cm, g[f"cands_d{depth}"]) and np.array_equal(fb, (act < 0).astype(np.int8))
    act2, cm2, fb2 = emu.greedy(sq, tm, mask=np.ascontiguousarray(g["mask"]), depth=depth)
    assert np.array_equal(act2, act) and np.array_equal(cm2, cm)
    # history guard: the chosen action among the agent's last three -> fallback, action -1
    hist = np.full((len(sq), 2, 3), -1, np.int8)
    hist[np.arange(len(sq)), tm, 1] = act.astype(np.int8)  # (-1 = "none" where nothing was chosen)
    act3, cm3, fb3 = emu.greedy(sq, tm, hist=hist, depth=depth)
    assert (fb3 == 1).all() and (act3 == -1).all() and np.array_equal(cm3, cm)
    o = oracle.batch_greedy(sq, tm, hist=hist, depth=depth)
    assert np.array_equal(o[0], act3) and np.array_equal(o[1], cm3) and np.array_equal(o[2], fb3)


def test_greedy_vs_oracle_selfplay(boards):
    state, tm, dn, rng = selfplay_states(6000, 26, seed=8)
    live = oracle.batch_winner(state) == 0
    state, tm = state[live], tm[live]
    # plus the dense random valid boards of the golden set (non-terminal ones), both movers
    dense = boards["squares"][boards["winner"] == 0]
    state = np.ascontiguousarray(np.concatenate([state, dense, dense]))
    tm = np.ascontiguousarray(np.concatenate([tm, np.zeros(len(dense), np.int8), np.ones(len(dense), np.int8)]))
    hist = rng.integers(-1, 54, (len(state), 2, 3)).astype(np.int8)
    for depth in (1, 2):
        for h in (None, hist):
            e = emu.greedy(state, tm, hist=h, depth=depth)
            o = oracle.batch_greedy(state, tm, hist=h, depth=depth)
            for x, y in zip(e, o):
                assert np.array_equal(x, y)
    # a mask that is a strict subset of the legal moves (the policy only searches the moves it is handed)
    m = oracle.batch_legal_mask(state, tm) * (rng.random((len(state), 54)) < 0.7)
    m = np.ascontiguousarray(m.astype(np.int8))
    e = emu.greedy(state, tm, mask=m, depth=2)
    o = oracle.batch_greedy(state, tm, mask=m, depth=2)
    for x, y in zip(e, o):
        assert np.array_equal(x, y)
    # (inside the pooled flow the closed form of the depth-2 loop, greedy_replay_closed, is cross-checked against the
    #  loop form, greedy_replay_sets, on every board; and EVERY placement from hand that the root rule settles without an
    #  evaluation -- its summary looked up in the board's table and its bits in the merged candidate sets -- against its
    #  exact evaluation)
    #  exact evaluation; and the FAST form of the pair evaluation -- threat squares, where reply_is_plain holds -- against the
    #  ordered line steps on every pair)
    pairs, not_plain, bad, settled, items = emu.greedy_stats()
    assert bad == 0
    print(f"pairs {pairs}, of them not plain {not_plain} ({not_plain / pairs:.4f})")
    assert 0.02 * pairs < not_plain < 0.2 * pairs  # both forms are exercised
    assert settled > 1.2 * pairs and 0 < items < pairs  # the rule settles most candidates, from a few items per board
    # the per-board composition (greedy_decide) next to the kernel's pooled one
    for kw in ({"hist": hist}, {"mask": m}):
        for x, y in zip(emu.greedy(state, tm, depth=2, pooled=False, **kw), oracle.batch_greedy(state, tm, depth=2, **kw)):
            assert np.array_equal(x, y)


def test_c1_thousand_reference_games(golden_dir):
    from tests.test_oracle_golden import c1_plies
    g, before, mover, mask = c1_plies(golden_dir)
    assert np.array_equal(emu.legal_mask(before, mover), mask)
    state, tm, dn = before.copy(), mover.copy(), np.zeros(len(mover), np.int8)
    out = emu.step(state, tm, dn, g["action"].astype(np.int32))
    assert np.array_equal(state, g["squares_after"]) and np.array_equal(out["winner"], g["winner"])


def test_validate_flags(boards):
    sq = np.ascontiguousarray(boards["squares"][:100]).copy()
    assert (emu.validate(sq) == 0).all()
    sq[3, 0] = 5          # a large piece on the small level
    sq[7, 9:11] = (3, 3)  # piece 3 twice
    sq[9, 18] = 9         # not a piece at all
    f = emu.validate(sq)
    assert f[3] & 1 and f[7] & 2 and f[9] & 1 and (np.delete(f, [3, 7, 9]) == 0).all()
    with pytest.raises(Exception, match="PIECE HAS BEEN USED TWICE"):
        oracle.is_legal(sq[7], 18, 0)  # what the reference does on that board (board.py:94-95)


@pytest.mark.parametrize("depth", [1, 2])
@pytest.mark.parametrize("pooled", [True, False])
def test_greedy_restricted_masks(golden_dir, depth, pooled):
    g = np.load(os.path.join(golden_dir, "greedy_restricted.npz"))
    act, cm, fb = emu.greedy(np.ascontiguousarray(g["squares"]), np.ascontiguousarray(g["to_move"]),
                             mask=np.ascontiguousarray(g["mask"]), depth=depth, pooled=pooled)
    assert np.array_equal(act, g[f"chosen_d{depth}"].astype(np.int32))
    assert np.array_equal(cm, g[f"cands_d{depth}"])
    assert np.array_equal(fb, (g[f"chosen_d{depth}"] < 0).astype(np.int8))


def test_greedy_depth3(golden_dir):
    """depth 3: device code == the oracle's literal restatement of greedy_policy.py:160-208, every position,
    with and without history; and == the reference's depth-3 decisions on the sampled positions."""
    d3 = np.load(os.path.join(golden_dir, "greedy_depth3.npz"))
    rng = np.random.default_rng(3)
    for tag, name in (("full", "greedy.npz"), ("restricted", "greedy_restricted.npz")):
        g = np.load(os.path.join(golden_dir, name))
        sq, tm, m = (np.ascontiguousarray(g[k]) for k in ("squares", "to_move", "mask"))
        hist = rng.integers(-1, 54, size=(len(sq), 2, 3)).astype(np.int8)
        for h in (None, hist):
            e = emu.greedy(sq, tm, mask=m, hist=h, depth=3)
            o = oracle.batch_greedy(sq, tm, mask=m, hist=h, depth=3)
            assert all(np.array_equal(a, b) for a, b in zip(e, o))
        idx = d3[f"index_{tag}"]
        act, cm, _ = emu.greedy(sq, tm, mask=m, depth=3)
        assert np.array_equal(act[idx], d3[f"chosen_d3_{tag}"].astype(np.int32))
        assert np.array_equal(cm[idx], d3[f"cands_d3_{tag}"])


def test_winner_exhaustive_tops():
    """check_for_winner over every pattern of tops (3^9, at each level) and 100k random stacks: the device
    code finds the highest-index complete line with packed arithmetic, the oracle walks the 8 lines."""
    import itertools
    cfg = np.array(list(itertools.product((0, 1, -1), repeat=9)), np.int8)
    for lvl in range(3):
        st = np.zeros((len(cfg), 27), np.int8)
        st[:, 9 * lvl:9 * lvl + 9] = cfg * (2 * lvl + 1)
        assert np.array_equal(emu.winner(np.ascontiguousarray(st)), oracle.batch_winner(st)), lvl
    rng = np.random.default_rng(1)
    st = np.zeros((100000, 27), np.int8)
    for lvl, vals in enumerate(((1, 2), (3, 4), (5, 6))):
        occ = rng.random((len(st), 9)) < 0.45
        v = rng.choice(vals, size=(len(st), 9)) * rng.choice((1, -1), size=(len(st), 9))
        st[:, 9 * lvl:9 * lvl + 9] = np.where(occ, v, 0)
    w = oracle.batch_winner(st)
    assert np.array_equal(emu.winner(st), w) and len(np.unique(w)) == 3


def test_greedy_policy_step_selfplay():
    """gbl_greedy_act's flow (decision + fallback draw + history append) against the oracle's, as two greedy
    agents play 24 plies on 1500 boards: the histories fill up, so the 3-move repeat guard and the fallback
    draw both fire."""
    n, seed, base = 1500, 5, 77
    s1, t1, d1 = oracle.batch_reset(n)
    h_emu = np.full((n, 2, 3), -1, np.int8); h_ora = h_emu.copy()
    fallbacks = 0
    for call in range(24):
        depth = 2 if call % 3 else 1
        e = emu.greedy_act(s1, t1, h_emu, seed, base, call, depth=depth)
        o = oracle.batch_greedy_act(s1, t1, h_ora, seed, base, call, depth=depth)
        for x, y in zip(e, o):
            assert np.array_equal(x, y), call
        assert np.array_equal(h_emu, h_ora)
        fallbacks += int(o[3].sum())
        a = np.where(d1 != 0, 0, o[0]).astype(np.int32)
        oracle.batch_step(s1, t1, d1, a, auto_reset=True)
    assert fallbacks > 100 and (h_ora >= 0).all()


def test_greedy_root_rule(boards):
    """The root rule in its LOOP form (tests/emu/greedy_root_rule.h; the kernel's table form is checked on every settled
    candidate inside emu.greedy's pooled flow, test_greedy_vs_oracle_selfplay): the summaries of our placements from hand,
    derived from the opponent's winning moves on the ROOT alone, equal the exact depth-2 evaluation on every candidate --
    selfplay positions, the dense random boards of the golden set (both movers), restricted masks."""
    state, tm, dn, rng = selfplay_states(8000, 30, seed=11)
    live = oracle.batch_winner(state) == 0
    dense = boards["squares"][boards["winner"] == 0]
    state = np.ascontiguousarray(np.concatenate([state[live], dense, dense]))
    tm = np.ascontiguousarray(np.concatenate([tm[live], np.zeros(len(dense), np.int8), np.ones(len(dense), np.int8)]))
    n_boards, settled, threatened, left, bad = emu.greedy_root_rule(state, tm)
    assert bad == 0 and n_boards == len(state) and settled > 8 * n_boards and threatened > 0.2 * settled and left > 0
    m = np.ascontiguousarray((oracle.batch_legal_mask(state, tm) * (rng.random((len(state), 54)) < 0.6)).astype(np.int8))
    assert emu.greedy_root_rule(state, tm, mask=m)[4] == 0


def test_greedy_virtual_root_rule(boards):
    """The root rule applied to moves of PLACED pieces through a virtual root (the position with the piece lifted; proven and
    measured in round 3, not shipped: tests/emu/greedy_root_rule.h): every candidate it settles equals the exact evaluation."""
    state, tm, dn, rng = selfplay_states(6000, 30, seed=12)
    live = oracle.batch_winner(state) == 0
    dense = boards["squares"][boards["winner"] == 0]
    state = np.ascontiguousarray(np.concatenate([state[live], dense, dense]))
    tm = np.ascontiguousarray(np.concatenate([tm[live], np.zeros(len(dense), np.int8), np.ones(len(dense), np.int8)]))
    for cap in (1, 6):
        n_boards, pairs, placed, roots, settled, left, items, bad = emu.greedy_vroot_rule(state, tm, cap=cap)
        assert bad == 0 and n_boards == len(state) and 0 < settled <= placed <= pairs and roots > 0, cap
    m = np.ascontiguousarray((oracle.batch_legal_mask(state, tm) * (rng.random((len(state), 54)) < 0.6)).astype(np.int8))
    assert emu.greedy_vroot_rule(state, tm, mask=m)[7] == 0


def test_greedy_on_terminal_roots(boards):
    """The root analysis takes the QUIET form of outcomes54 unless a board of the wavefront holds a line (outcomes54_root):
    roots that do, both movers, depth 1 and 2, pooled flow and per-board composition, against the oracle."""
    sq = terminal_roots(boards)
    assert len(sq) > 2000
    for me in (0, 1):
        tm = np.full(len(sq), me, np.int8)
        for depth in (1, 2):
            o = oracle.batch_greedy(sq, tm, depth=depth)
            for pooled in (True, False):
                for x, y in zip(emu.greedy(sq, tm, depth=depth, pooled=pooled), o):
                    assert np.array_equal(x, y), (me, depth, pooled)
    assert emu.greedy_stats()[2] == 0
