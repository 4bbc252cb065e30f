"""The CPU oracle (oracle/gobblet_oracle.c) against the golden vectors produced by
running the reference (tests/golden/make_golden.py) and against the upstream
known-answer test (reference tests/test_manual_policy_collector.py:49-507)."""
import json
import os

import numpy as np
import pytest

import oracle


@pytest.fixture(scope="module")
def kat(golden_dir):
    return json.load(open(os.path.join(golden_dir, "kat_collector.json")))


@pytest.fixture(scope="module")
def games(golden_dir):
    return np.load(os.path.join(golden_dir, "random_games.npz"))


@pytest.fixture(scope="module")
def boards(golden_dir):
    return np.load(os.path.join(golden_dir, "board_functions.npz"))


@pytest.fixture(scope="module")
def greedy(golden_dir):
    return np.load(os.path.join(golden_dir, "greedy.npz"))


def test_reset_starting():
    # reference tests/test_gobblet_env.py:23-28
    state, to_move, done = oracle.batch_reset(3)
    assert (state == np.zeros(27)).all() and (to_move == 0).all() and (done == 0).all()


def test_upstream_kat_masks_and_board(kat):
    """Fixed sequence 18, 36, 28, 46 then illegal 29 (test_manual_policy_collector.py)."""
    assert all(kat["reference_reproduces"].values())  # the reference itself reproduces its literals
    s = np.zeros(27, np.int8); tm, dn = 0, 0
    assert oracle.legal_mask(s, tm).tolist() == kat["mask_after"]["output0"]
    for name, a in zip(["output1", "output2", "output3", "output4"], kat["actions"]):
        s, tm, dn, w, rw = oracle.step(s, tm, dn, a)
        assert w == 0 and dn == 0
        assert oracle.legal_mask(s, tm).tolist() == kat["mask_after"][name], name
    assert oracle.legal_mask(s, tm).tolist() == kat["mask_after"]["output5"]
    assert np.flatnonzero(oracle.legal_mask(s, tm)).tolist() == kat["legal_moves_output6"]
    assert not oracle.is_legal(s, kat["illegal_action"], tm)
    s2, tm2, dn2, w, rw = oracle.step(s, tm, dn, kat["illegal_action"])
    assert s2.tolist() == kat["board_output8"]            # board unchanged
    assert tm2 == kat["reference_to_move_after_illegal"]   # ... but the turn passes (gobblet.py:244-246)
    assert dn2 == 0 and rw.tolist() == [0, 0]


def test_board_functions(boards):
    sq = boards["squares"]
    assert np.array_equal(oracle.batch_flatboard(sq), boards["flatboard"])
    assert np.array_equal(oracle.batch_covered(sq), boards["covered"])
    assert np.array_equal(oracle.batch_winner(sq), boards["winner"])
    n = len(sq)
    assert np.array_equal(oracle.batch_legal_mask(sq, np.zeros(n, np.int8)), boards["legal_p1"])
    assert np.array_equal(oracle.batch_legal_mask(sq, np.ones(n, np.int8)), boards["legal_p2"])
    assert np.array_equal(oracle.batch_observe(sq, np.zeros(n, np.int8), 0), boards["obs_p1"])
    assert np.array_equal(oracle.batch_observe(sq, np.zeros(n, np.int8), 1), boards["obs_p2"])
    for i in range(n):
        assert oracle.check_game_over(sq[i]) == bool(boards["game_over"][i])
        for p in range(0, 9, 4):
            for s in (1, 2, 3):
                assert oracle.get_action(sq[i], p, s, 0) == boards["get_action_p1"][i][p][s - 1]
                assert oracle.get_action(sq[i], p, s, 1) == boards["get_action_p2"][i][p][s - 1]


def test_edge_cases_named(boards):
    names = [str(x) for x in boards["edge_names"]]
    w = dict(zip(names, boards["winner"][:len(names)].tolist()))
    # the LAST matching line decides
    assert w["dual_p1_line0_p2_line2"] == -1 and w["dual_p2_line0_p1_line2"] == 1
    for i, k in enumerate(names):
        assert oracle.check_for_winner(boards["squares"][i]) == w[k]
    # covered pieces generate no legal action for that piece number (App. D.4)
    i = names.index("covered_stack")
    m = oracle.legal_mask(boards["squares"][i], 0)
    assert m[0:9].sum() == 0        # P1 small piece 1 under a medium+large
    assert m[9:18].sum() == 0       # P1 small piece 2 under its own medium (own pieces cover too, board.py:219)
    assert m[27:36].sum() > 0       # P1 medium piece 4 sits on top -> movable
    m2 = oracle.legal_mask(boards["squares"][i], 1)
    assert m2[9:18].sum() == 0 and m2[18:27].sum() == 0  # P2 small 2 under a large, P2 medium 3 under a large
    # self-gobble allowed, same-square "move" not (App. D.3)
    i = names.index("self_gobble")
    m = oracle.legal_mask(boards["squares"][i], 0)
    assert m[0:9].sum() == 0 and m[18 + 4] == 0 and m[36 + 4] == 1


def test_random_games_step_by_step(games):
    g = games
    n = len(g["action"])
    for i in range(n):
        if g["ply"][i] == 0:
            s = np.zeros(27, np.int8); tm, dn = 0, 0
        assert np.array_equal(s, g["squares_before"][i])
        assert tm == g["mover"][i]
        assert int(oracle.is_legal(s, g["action"][i], tm)) == g["legal_before"][i]
        s, tm, dn, w, rw = oracle.step(s, tm, dn, int(g["action"][i]))
        assert np.array_equal(s, g["squares_after"][i]), i
        assert tm == g["to_move_after"][i]
        assert w == g["winner"][i] and dn == g["done"][i]
        assert rw.tolist() == g["reward"][i].tolist()
        o = oracle.observe(s, tm, tm)
        assert np.array_equal(o["action_mask"], g["mask_next"][i])
        assert np.array_equal(o["observation"], (g["obs_p1"], g["obs_p2"])[tm][i])
        off = oracle.observe(s, 1 - tm, tm)
        assert np.array_equal(off["action_mask"], g["mask_offturn"][i]) and off["action_mask"].sum() == 0
        assert np.array_equal(off["observation"], (g["obs_p1"], g["obs_p2"])[1 - tm][i])


def test_random_games_batched(games):
    """The same plies through the batched driver (the arrays the HIP C-ABI uses)."""
    g = games
    state = g["squares_before"].copy(); tm = g["mover"].copy(); dn = np.zeros(len(tm), np.int8)
    out = oracle.batch_step(state, tm, dn, g["action"])
    assert np.array_equal(state, g["squares_after"]) and np.array_equal(tm, g["to_move_after"])
    assert np.array_equal(out["winner"], g["winner"]) and np.array_equal(dn, g["done"])
    assert np.array_equal(out["reward"], g["reward"])
    live = g["done"] == 0
    assert np.array_equal(out["mask"][live], g["mask_next"][live])
    assert (out["mask"][~live] == 0).all()  # frozen boards: nobody is to move
    obs_next = np.where(g["to_move_after"][:, None, None, None] == 0, g["obs_p1"], g["obs_p2"])
    assert np.array_equal(out["obs"], obs_next)
    # multi-threaded driver = same bytes
    state2 = g["squares_before"].copy(); tm2 = g["mover"].copy(); dn2 = np.zeros(len(tm), np.int8)
    out2 = oracle.batch_step(state2, tm2, dn2, g["action"], threads=4)
    for k in out:
        assert np.array_equal(out[k], out2[k])
    assert np.array_equal(state, state2)


def test_illegal_terminate_mode(games):
    """env() layer: TerminateIllegalWrapper(illegal_reward=-1) (gobblet.py:114, docstring :50-51).
    Restated from the call site; the wrapper itself is third-party (parity unpinned here)."""
    g = games
    ill = np.flatnonzero(g["legal_before"] == 0)
    assert len(ill) > 5
    for i in ill[:20]:
        s, tm, dn, w, rw = oracle.step(g["squares_before"][i], int(g["mover"][i]), 0, int(g["action"][i]),
                                       illegal_mode=oracle.ILLEGAL_TERMINATE)
        assert np.array_equal(s, g["squares_before"][i]) and dn == 1 and w == 0
        assert rw[g["mover"][i]] == -1 and rw[1 - g["mover"][i]] == 0


def test_greedy_decode_obs(greedy):
    for i in range(0, len(greedy["squares"]), 3):
        s, agent = oracle.greedy_decode_obs(greedy["obs"][i])
        assert agent == greedy["to_move"][i] and np.array_equal(s, greedy["squares"][i])


@pytest.mark.parametrize("depth", [1, 2])
def test_greedy_decisions(greedy, depth):
    g = greedy
    n = len(g["squares"])
    for i in range(n):
        chosen, cands, fb = oracle.greedy(g["squares"][i], int(g["to_move"][i]), g["mask"][i], depth=depth)
        ref_chosen = int(g[f"chosen_d{depth}"][i])
        assert (chosen if chosen is not None else -1) == ref_chosen, i
        assert np.array_equal(cands, g[f"cands_d{depth}"][i]), i
        assert fb == (ref_chosen < 0)
        if chosen is not None:  # history guard (greedy_policy.py:211-214)
            _, _, fb2 = oracle.greedy(g["squares"][i], int(g["to_move"][i]), g["mask"][i], depth=depth,
                                      prev3=[50, chosen])
            assert fb2
    act, cm, fb = oracle.batch_greedy(g["squares"], g["to_move"], depth=depth)
    ref = g[f"chosen_d{depth}"].astype(np.int32)
    assert np.array_equal(act, ref) and np.array_equal(cm, g[f"cands_d{depth}"])


def test_greedy_immediate_win_overwritten(greedy):
    """depth 1 returns the winning move, depth 2 overwrites it."""
    i = 320
    assert int(greedy["chosen_d1"][i]) == 8 and int(greedy["chosen_d2"][i]) == 7
    assert oracle.greedy(greedy["squares"][i], 0, greedy["mask"][i], depth=1)[0] == 8
    assert oracle.greedy(greedy["squares"][i], 0, greedy["mask"][i], depth=2)[0] == 7


def test_philox_known_answers():
    """Random123 kat_vectors for philox4x32-10."""
    f = oracle.philox4x32_10
    assert [hex(x) for x in f([0, 0, 0, 0], [0, 0])] == ["0x6627e8d5", "0xe169c58d", "0xbc57ac4c", "0x9b00dbd8"]
    assert [hex(x) for x in f([0xFFFFFFFF] * 4, [0xFFFFFFFF] * 2)] == ["0x408f276d", "0x41c83b0e", "0xa20bc7c6",
                                                                      "0x6d5451fd"]
    assert [hex(x) for x in f([0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344], [0xA4093822, 0x299F31D0])] == [
        "0xd16cfe09", "0x94fdcceb", "0x5001e420", "0x24126ea1"]


def test_sampler_uniform_over_legal():
    mask = np.zeros(54, np.int8); mask[[3, 7, 20, 53]] = 1
    draws = np.array([oracle.sample_action(mask, 0, e, 5) for e in range(4000)])
    assert set(draws.tolist()) == {3, 7, 20, 53}
    counts = np.bincount(draws, minlength=54)[[3, 7, 20, 53]]
    assert counts.min() > 850 and counts.max() < 1150
    assert oracle.sample_action(np.zeros(54, np.int8), 0, 0, 0) == -1
    assert np.array_equal(oracle.batch_sample(np.tile(mask, (64, 1)), 0, 100, 5), draws[100:164])


def test_rollout_equals_sample_then_step():
    """Fused rollout == (mask -> sample -> step with auto-reset) ply by ply; thread-count invariant."""
    n, plies, seed = 512, 40, 3
    s1, t1, d1 = oracle.batch_reset(n)
    r = oracle.batch_rollout(s1, t1, d1, seed, 1000, 0, plies, threads=3)
    s2, t2, d2 = oracle.batch_reset(n)
    games = 0
    for t in range(plies):
        m = oracle.batch_legal_mask(s2, t2)
        a = oracle.batch_sample(m, seed, 1000, t)
        out = oracle.batch_step(s2, t2, d2, a, auto_reset=True)
        games += int(d2.sum())
    assert np.array_equal(s1, s2) and np.array_equal(t1, t2) and np.array_equal(d1, d2)
    assert np.array_equal(r["mask"], out["mask"]) and np.array_equal(r["obs"], out["obs"])
    assert np.array_equal(r["winner"], out["winner"]) and np.array_equal(r["actions"], a)
    assert r["counters"][0] == n * plies and r["counters"][1] == games and games > n
    assert r["counters"][2] + r["counters"][3] == games


def c1_plies(golden_dir):
    """The C1 fixture (1000 reference games) flattened to independent plies: (before, mover, action, after, mask)."""
    g = np.load(os.path.join(golden_dir, "c1_1000_games.npz"))
    n = len(g["action"])
    starts = np.concatenate([[0], np.cumsum(g["game_len"])[:-1]])
    first = np.zeros(n, bool); first[starts] = True
    before = np.zeros((n, 27), np.int8)
    before[~first] = g["squares_after"][:-1][~first[1:]]
    ply_in_game = np.arange(n) - np.repeat(starts, g["game_len"])
    mover = (ply_in_game % 2).astype(np.int8)
    mask = np.unpackbits(g["mask_before"], axis=1)[:, :54].astype(np.int8)
    return g, before, mover, mask


def test_c1_thousand_reference_games(golden_dir):
    """1000 masked-random games played by the reference; every ply's mask, board and
    winner through the oracle's batched step."""
    g, before, mover, mask = c1_plies(golden_dir)
    assert np.array_equal(oracle.batch_legal_mask(before, mover), mask)
    state, tm, dn = before.copy(), mover.copy(), np.zeros(len(mover), np.int8)
    out = oracle.batch_step(state, tm, dn, g["action"].astype(np.int32), threads=4)
    assert np.array_equal(state, g["squares_after"]) and np.array_equal(out["winner"], g["winner"])
    ends = np.cumsum(g["game_len"]) - 1
    assert (g["winner"][ends] != 0).all() and (np.delete(g["winner"], ends) == 0).all()
    assert np.array_equal(dn, (g["winner"] != 0).astype(np.int8))


@pytest.mark.parametrize("depth", [1, 2])
def test_greedy_restricted_masks(golden_dir, depth):
    """Subsets of the legal moves (1-3 actions / random 40 %): the len(actions_depth1) > 1 guards and early
    breaks of greedy_policy.py:98-101,132-136."""
    g = np.load(os.path.join(golden_dir, "greedy_restricted.npz"))
    act, cm, fb = oracle.batch_greedy(np.ascontiguousarray(g["squares"]), np.ascontiguousarray(g["to_move"]),
                                      mask=np.ascontiguousarray(g["mask"]), depth=depth)
    assert np.array_equal(act, g[f"chosen_d{depth}"].astype(np.int32))
    assert np.array_equal(cm, g[f"cands_d{depth}"])
    assert np.array_equal(fb, (g[f"chosen_d{depth}"] < 0).astype(np.int8))


def test_greedy_depth3(golden_dir):
    """depth=3 of the reference (greedy_policy.py:160-208, restated literally in the oracle) on a sample
    of both greedy fixtures: the reference's own depth-3 decisions, which equal its depth-2 ones."""
    d3 = np.load(os.path.join(golden_dir, "greedy_depth3.npz"))
    for tag, name in (("full", "greedy.npz"), ("restricted", "greedy_restricted.npz")):
        g = np.load(os.path.join(golden_dir, name))
        idx = d3[f"index_{tag}"]
        assert len(idx) >= 100
        act, cm, fb = oracle.batch_greedy(np.ascontiguousarray(g["squares"][idx]), np.ascontiguousarray(g["to_move"][idx]),
                                          mask=np.ascontiguousarray(g["mask"][idx]), depth=3)
        assert np.array_equal(act, d3[f"chosen_d3_{tag}"].astype(np.int32))
        assert np.array_equal(cm, d3[f"cands_d3_{tag}"])
        assert np.array_equal(fb, (d3[f"chosen_d3_{tag}"] < 0).astype(np.int8))
        assert np.array_equal(d3[f"chosen_d3_{tag}"], g["chosen_d2"][idx])  # the block has no observable effect
        assert np.array_equal(d3[f"cands_d3_{tag}"], g["cands_d2"][idx])


def test_turn_counter(games):
    """raw_env.turn (gobblet.py:270,289): +1 whenever raw step runs, illegal no-ops included."""
    g = games
    state = g["squares_before"].copy(); tm = g["mover"].copy(); dn = np.zeros(len(tm), np.int8)
    turn = g["ply"].astype(np.int32).copy()
    oracle.batch_step(state, tm, dn, g["action"], turn=turn)
    assert np.array_equal(turn, g["ply"] + 1)
