"""GPU parity tests (-m gpu): the HIP path, called through the Python host code and the C-ABI, against
the oracle on the same seeded inputs, against the committed golden vectors, and -- at
BASELINE.json's full sizes -- through size-independent properties.  Bit-exact everywhere (all
arithmetic is integer / byte work)."""
import json
import os

import numpy as np
import pytest
import torch

import oracle

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


@pytest.fixture(scope="module")
def G():
    import gobblet_rl_amd as g
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    g._native.lib()
    return g


def t(a, dtype=None):
    x = torch.from_numpy(np.ascontiguousarray(a)).to(DEV)
    return x if dtype is None else x.to(dtype)


def npy(x):
    return x.cpu().numpy()


def selfplay(n, plies, seed):
    """n boards in staggered game phases, via the oracle."""
    state, tm, dn = oracle.batch_reset(n)
    for k in range(plies):
        m = oracle.batch_legal_mask(state, tm)
        a = oracle.batch_sample(m, seed, 0, k)
        live = (np.arange(n) % plies) > k
        s2, t2, d2 = state.copy(), tm.copy(), dn.copy()
        oracle.batch_step(s2, t2, d2, a, threads=8)
        state[live], tm[live], dn[live] = s2[live], t2[live], d2[live]
    return state, tm, dn


def vec_env(G, n, state, tm, dn, **kw):
    env = G.BatchedGobblet(n, DEV, **kw)
    env.board.squares = t(state)
    env.to_move.copy_(t(tm)); env.done.copy_(t(dn))
    env.refresh()
    return env


# ---- Board-level functions vs the golden vectors (reference outputs) ---------------------------------
@pytest.mark.parametrize("n", [1, 3, 63, 64, 65, 408])
def test_board_functions_vs_golden(G, golden_dir, n):
    g = np.load(os.path.join(golden_dir, "board_functions.npz"))
    b = G.BatchedBoard(n, DEV, squares=t(g["squares"][:n]))
    assert np.array_equal(npy(b.get_flatboard()), g["flatboard"][:n])
    assert np.array_equal(npy(b.check_covered()), g["covered"][:n])
    assert np.array_equal(npy(b.check_for_winner()), g["winner"][:n])
    assert np.array_equal(npy(b.check_game_over()), g["game_over"][:n].astype(bool))
    assert np.array_equal(npy(b.legal_mask(0)), g["legal_p1"][:n])
    assert np.array_equal(npy(b.legal_mask(1)), g["legal_p2"][:n])
    assert np.array_equal(npy(b.observation(0)), g["obs_p1"][:n])
    assert np.array_equal(npy(b.observation(1)), g["obs_p2"][:n])
    for a in (0, 13, 26, 40, 53):
        assert np.array_equal(npy(b.is_legal(a, 0)), g["legal_p1"][:n, a].astype(bool))
        assert np.array_equal(npy(b.is_legal(a, 1)), g["legal_p2"][:n, a].astype(bool))
    for p in (0, 4, 8):
        for s in (1, 2, 3):
            assert np.array_equal(npy(b.get_action(p, s, 0)), g["get_action_p1"][:n, p, s - 1])
            assert np.array_equal(npy(b.get_action(p, s, 1)), g["get_action_p2"][:n, p, s - 1])


def test_upstream_kat(G, golden_dir):
    """reference tests/test_manual_policy_collector.py: 18, 36, 28, 46 then illegal 29."""
    kat = json.load(open(os.path.join(golden_dir, "kat_collector.json")))
    env = G.BatchedGobblet(1, DEV)
    assert npy(env.action_mask)[0].tolist() == kat["mask_after"]["output0"]
    assert (npy(env.squares) == 0).all()  # tests/test_gobblet_env.py:23-28
    for name, a in zip(["output1", "output2", "output3", "output4"], kat["actions"]):
        obs, rew, done, win = env.step([a])
        assert npy(obs["action_mask"])[0].tolist() == kat["mask_after"][name]
        assert int(done[0]) == 0 and int(win[0]) == 0
    assert np.flatnonzero(npy(env.action_mask)[0]).tolist() == kat["legal_moves_output6"]
    env.step([kat["illegal_action"]])
    assert npy(env.squares)[0].tolist() == kat["board_output8"]
    assert int(env.to_move[0]) == kat["reference_to_move_after_illegal"]


def test_step_vs_golden_games(G, golden_dir):
    g = np.load(os.path.join(golden_dir, "random_games.npz"))
    n = len(g["action"])
    env = vec_env(G, n, g["squares_before"], g["mover"], np.zeros(n, np.int8))
    obs, rew, done, win = env.step(t(g["action"]))
    assert np.array_equal(npy(env.squares), g["squares_after"])
    assert np.array_equal(npy(env.to_move), g["to_move_after"])
    assert np.array_equal(npy(win), g["winner"]) and np.array_equal(npy(done), g["done"])
    assert np.array_equal(npy(rew), g["reward"])
    live = g["done"] == 0
    m = npy(obs["action_mask"])
    assert np.array_equal(m[live], g["mask_next"][live]) and (m[~live] == 0).all()
    obs_next = np.where(g["to_move_after"][:, None, None, None] == 0, g["obs_p1"], g["obs_p2"])
    assert np.array_equal(npy(obs["observation"]), obs_next)


# ---- fused step vs the oracle, all modes, ragged and large sizes ---------------------------------------
@pytest.mark.parametrize("n,illegal,auto_reset,with_obs", [
    (4096, "noop", False, True), (4096, "noop", True, True), (4099, "terminate", False, True),
    (1000, "terminate", True, False), (37, "noop", False, False), (262144, "noop", True, True)])
def test_step_vs_oracle(G, n, illegal, auto_reset, with_obs):
    rng = np.random.default_rng(n)
    state, tm, dn = selfplay(n, 24, seed=n % 1000)
    env = vec_env(G, n, state, tm, dn, illegal_mode=illegal, auto_reset=auto_reset, with_observation=with_obs)
    imode = {"noop": 0, "terminate": 1}[illegal]
    status = torch.full((n,), 77, dtype=torch.int8, device=DEV)
    nxt = torch.full((n,), 77, dtype=torch.int32, device=DEV)
    for k in range(4 if n > 100000 else 8):
        m = oracle.batch_legal_mask(state, tm)
        a = oracle.batch_sample(m, 5, 0, 100 + k)
        wild = rng.random(n) < 0.12
        a = np.where(wild, rng.integers(-3, 58, n), a).astype(np.int32)
        exp_status = oracle.batch_action_status(state, tm, dn, a, auto_reset=auto_reset)   # (of the position BEFORE the step)
        o = oracle.batch_step(state, tm, dn, a, illegal_mode=imode, auto_reset=auto_reset, threads=8)
        if k % 2 == 0:
            obs, rew, done, win = env.step(t(a))
        else:
            # gbl_step_ex (k_step<..., EXT>): the status byte of every action (out-of-range = illegal + flagged) and
            # the next mover's masked-uniform draw from the mask this launch stores (= gbl_sample behind the step; -1 where
            # nobody is to move: a frozen board)
            obs, rew, done, win = env.step(t(a), status=status, next_actions=nxt)
            assert np.array_equal(npy(status), exp_status), k
            assert set(np.unique(exp_status)) <= {0, 1, 3} and ((exp_status == 3).any() or n < 1000)
            assert np.array_equal(npy(nxt), oracle.batch_sample(o["mask"], env.seed, env.env_base, k + 1)), k
        assert np.array_equal(npy(env.squares), state), k
        assert np.array_equal(npy(env.to_move), tm) and np.array_equal(npy(done), dn)
        assert np.array_equal(npy(win), o["winner"]) and np.array_equal(npy(rew), o["reward"])
        assert np.array_equal(npy(obs["action_mask"]), o["mask"])
        if with_obs:
            assert np.array_equal(npy(obs["observation"]), o["obs"])
        else:
            assert obs["observation"] is None


def test_step_optional_outputs_null(G):
    """winner / reward / mask / obs pointers may be NULL (MASK_ONLY and bare variants)."""
    from gobblet_rl_amd import _native as nat
    n = 1000
    state, tm, dn = selfplay(n, 20, seed=3)
    a = oracle.batch_sample(oracle.batch_legal_mask(state, tm), 1, 0, 0)
    exp_s, exp_t, exp_d = state.copy(), tm.copy(), dn.copy()
    oracle.batch_step(exp_s, exp_t, exp_d, a)
    s, tmv, d, av = t(state), t(tm), t(dn), t(a)
    nat.check(nat.lib().gbl_step(s.data_ptr(), tmv.data_ptr(), d.data_ptr(), av.data_ptr(), None, None, None, None,
                                 None, n, 0, 0, None))
    torch.cuda.synchronize()
    assert np.array_equal(npy(s), exp_s) and np.array_equal(npy(tmv), exp_t) and np.array_equal(npy(d), exp_d)


def test_board_api_vs_oracle(G):
    n = 5000
    rng = np.random.default_rng(1)
    state, tm, dn = selfplay(n, 30, seed=11)
    b = G.BatchedBoard(n, DEV, squares=t(state))
    assert np.array_equal(npy(b.legal_mask(t(tm))), oracle.batch_legal_mask(state, tm))
    assert np.array_equal(npy(b.check_for_winner()), oracle.batch_winner(state))
    assert np.array_equal(npy(b.get_flatboard()), oracle.batch_flatboard(state))
    assert np.array_equal(npy(b.check_covered()), oracle.batch_covered(state))
    assert np.array_equal(npy(b.observation(t(tm))), oracle.batch_observe(state, tm, -1))
    a = rng.integers(-2, 56, n).astype(np.int32)
    ag = rng.integers(0, 2, n).astype(np.int8)
    exp = np.array([(0 <= a[i] < 54) and oracle.is_legal(state[i], a[i], ag[i]) for i in range(n)])
    assert np.array_equal(npy(b.is_legal(t(a), t(ag))), exp)
    b.play_turn(t(ag), t(a))
    exp_s = np.stack([oracle.play_turn(state[i], ag[i], a[i]) if 0 <= a[i] < 54 else state[i] for i in range(n)])
    assert np.array_equal(npy(b.squares), exp_s)


@pytest.mark.parametrize("n", [1, 63, 65, 4099])
def test_board_eval_vs_oracle(G, n):
    """gbl_board_eval: optional play_turn + everything the reference derives from the position, one launch."""
    from gobblet_rl_amd import _native as nat
    rng = np.random.default_rng(n)
    state, tm, dn = selfplay(n, 30, seed=3 + n)

    def check(rec, st):
        assert np.array_equal(npy(rec["squares"]), st)
        assert np.array_equal(npy(rec["winner"]), oracle.batch_winner(st))
        assert np.array_equal(npy(rec["flat"]), oracle.batch_flatboard(st))
        assert np.array_equal(npy(rec["covered"]), oracle.batch_covered(st))
        for ag in (0, 1):
            who = np.full(n, ag, np.int8)
            assert np.array_equal(npy(rec["mask%d" % ag]), oracle.batch_legal_mask(st, who))
            assert np.array_equal(npy(rec["obs%d" % ag]), oracle.batch_observe(st, who, ag))
        raw = npy(rec["record"])
        used = np.zeros(nat.REC_BYTES, bool)
        for o, size in nat.REC_FIELDS.values():
            used[o:o + size] = True
        assert (raw[:, ~used] == 0).all()  # padding bytes are zero
    b = G.BatchedBoard(n, DEV, squares=t(state))
    check(b.evaluate(), state)
    a = rng.integers(-2, 56, n).astype(np.int32)
    legal = oracle.batch_sample(oracle.batch_legal_mask(state, tm), 5, 0, 0)
    a[::2] = legal[::2]  # half of the boards play a legal move of the agent to move
    exp_s = np.stack([oracle.play_turn(state[i], tm[i], a[i]) if 0 <= a[i] < 54 else state[i] for i in range(n)])
    check(b.evaluate(t(tm), t(a)), exp_s)
    assert np.array_equal(npy(b.squares), exp_s)  # the state is updated in place


def test_sampler_and_rollout_vs_oracle(G):
    n, plies, seed, base = 4096 + 17, 48, 9, 123456789012
    env = G.BatchedGobblet(n, DEV, auto_reset=True, seed=seed, env_base=base)
    s, tm, dn = oracle.batch_reset(n)
    # separate kernels: sample -> step, ply by ply
    for k in range(12):
        a = env.sample_actions()
        m = oracle.batch_legal_mask(s, tm)
        exp_a = oracle.batch_sample(m, seed, base, k)
        assert np.array_equal(npy(a), exp_a)
        env.step(a)
        oracle.batch_step(s, tm, dn, exp_a, auto_reset=True)
        assert np.array_equal(npy(env.squares), s)
    # board ids across the 2^32 boundary inside one wavefront (boards beyond it draw under a derived key)
    from gobblet_rl_amd import _native as nat
    m = oracle.batch_legal_mask(s, tm)
    md, act = t(m), torch.empty(n, dtype=torch.int32, device=DEV)
    for b0 in ((1 << 32) - 30, (3 << 32) - 2000):
        nat.check(nat.lib().gbl_sample(md.data_ptr(), act.data_ptr(), n, 2**63 + 11, b0, 4000000000, None))
        torch.cuda.synchronize()
        assert np.array_equal(npy(act), oracle.batch_sample(m, 2**63 + 11, b0, 4000000000))
    # fused rollout continues the same stream (ply index carries on)
    for plies in (1, plies):
        o = oracle.batch_rollout(s, tm, dn, seed, base, env.ply, plies, threads=8)
        before = npy(env.counters).copy()
        obs, rew, done, win = env.rollout(plies, count=True)
        assert np.array_equal(npy(env.squares), s) and np.array_equal(npy(env.to_move), tm)
        assert np.array_equal(npy(done), dn) and np.array_equal(npy(win), o["winner"])
        assert np.array_equal(npy(rew), o["reward"]) and np.array_equal(npy(env.actions), o["actions"])
        assert np.array_equal(npy(obs["action_mask"]), o["mask"])
        assert np.array_equal(npy(obs["observation"]), o["obs"])
        assert np.array_equal(npy(env.counters) - before, o["counters"])


def test_decode_obs_and_greedy_vs_golden(G, golden_dir):
    from gobblet_rl_amd import _native as nat
    g = np.load(os.path.join(golden_dir, "greedy.npz"))
    n = len(g["squares"])
    L = nat.lib()
    obs = t(g["obs"]); st = torch.empty((n, 27), dtype=torch.int8, device=DEV); who = torch.empty(n, dtype=torch.int8, device=DEV)
    nat.check(L.gbl_decode_obs(obs.data_ptr(), st.data_ptr(), who.data_ptr(), n, None))
    torch.cuda.synchronize()
    assert np.array_equal(npy(st), g["squares"]) and np.array_equal(npy(who), g["to_move"])
    for depth in (1, 2):
        act = torch.empty(n, dtype=torch.int32, device=DEV); cm = torch.empty((n, 54), dtype=torch.int8, device=DEV)
        fb = torch.empty(n, dtype=torch.int8, device=DEV)
        for mask in (None, t(g["mask"])):
            nat.check(L.gbl_greedy(st.data_ptr(), who.data_ptr(), nat.ptr(mask), None, depth, act.data_ptr(),
                                   cm.data_ptr(), fb.data_ptr(), n, None))
            torch.cuda.synchronize()
            assert np.array_equal(npy(act), g[f"chosen_d{depth}"].astype(np.int32))
            assert np.array_equal(npy(cm), g[f"cands_d{depth}"])
            assert np.array_equal(npy(fb), (g[f"chosen_d{depth}"] < 0).astype(np.int8))
    assert int(g["chosen_d1"][320]) == 8 and int(npy(act)[320]) == 7  # App. B quirk: depth 2 overwrites the win


def test_greedy_restricted_masks(G, golden_dir):
    """Reference decisions when handed 1-3 (or a random 40 % of the) legal moves: guard / break paths."""
    from gobblet_rl_amd import _native as nat
    g = np.load(os.path.join(golden_dir, "greedy_restricted.npz"))
    n = len(g["squares"])
    st, who, m = t(g["squares"]), t(g["to_move"]), t(g["mask"])
    for depth in (1, 2):
        act = torch.empty(n, dtype=torch.int32, device=DEV); cm = torch.empty((n, 54), dtype=torch.int8, device=DEV)
        fb = torch.empty(n, dtype=torch.int8, device=DEV)
        nat.check(nat.lib().gbl_greedy(st.data_ptr(), who.data_ptr(), m.data_ptr(), None, depth, act.data_ptr(),
                                       cm.data_ptr(), fb.data_ptr(), n, None))
        torch.cuda.synchronize()
        assert np.array_equal(npy(act), g[f"chosen_d{depth}"].astype(np.int32))
        assert np.array_equal(npy(cm), g[f"cands_d{depth}"])
        assert np.array_equal(npy(fb), (g[f"chosen_d{depth}"] < 0).astype(np.int8))


def test_greedy_depth3(G, golden_dir):
    """depth 3 (greedy_policy.py:160-208): gbl_greedy == the reference's own depth-3 decisions on the
    sampled positions and == the oracle's literal restatement on all of them."""
    from gobblet_rl_amd import _native as nat
    d3 = np.load(os.path.join(golden_dir, "greedy_depth3.npz"))
    for tag, name in (("full", "greedy.npz"), ("restricted", "greedy_restricted.npz")):
        g = np.load(os.path.join(golden_dir, name))
        n = len(g["squares"])
        st, who, m = t(g["squares"]), t(g["to_move"]), t(g["mask"])
        act = torch.empty(n, dtype=torch.int32, device=DEV); cm = torch.empty((n, 54), dtype=torch.int8, device=DEV)
        fb = torch.empty(n, dtype=torch.int8, device=DEV)
        nat.check(nat.lib().gbl_greedy(st.data_ptr(), who.data_ptr(), m.data_ptr(), None, 3, act.data_ptr(),
                                       cm.data_ptr(), fb.data_ptr(), n, None))
        torch.cuda.synchronize()
        idx = d3[f"index_{tag}"]
        assert np.array_equal(npy(act)[idx], d3[f"chosen_d3_{tag}"].astype(np.int32))
        assert np.array_equal(npy(cm)[idx], d3[f"cands_d3_{tag}"])
        o = oracle.batch_greedy(np.ascontiguousarray(g["squares"]), np.ascontiguousarray(g["to_move"]),
                                mask=np.ascontiguousarray(g["mask"]), depth=3)
        assert np.array_equal(npy(act), o[0]) and np.array_equal(npy(cm), o[1]) and np.array_equal(npy(fb), o[2])
    pol = G.GreedyGobbletPolicy(depth=3, device=DEV)
    g = np.load(os.path.join(golden_dir, "greedy.npz"))
    a = pol.compute_actions(g["obs"], g["mask"])  # host class, depth=3: no fallback in this set (empty history)
    assert np.array_equal(npy(a), g["chosen_d2"].astype(np.int32)) and not npy(pol.last_fallback).any()


def test_greedy_policy_step_vs_oracle(G):
    """gbl_greedy_act (decision + fallback draw + history append in one launch) against the oracle while two
    greedy agents play 24 plies on 5000 boards; then the host class, which is built on it."""
    from gobblet_rl_amd import _native as nat
    n, seed, base = 5000, 5, 1 << 33
    s1, t1, d1 = oracle.batch_reset(n)
    h_ora = np.full((n, 2, 3), -1, np.int8)
    h_dev = t(h_ora)
    out = torch.empty(n, dtype=torch.int32, device=DEV); ch = torch.empty(n, dtype=torch.int32, device=DEV)
    cm = torch.empty((n, 54), dtype=torch.int8, device=DEV); fb = torch.empty(n, dtype=torch.int8, device=DEV)
    fallbacks = 0
    for call in range(24):
        depth = 2 if call % 3 else 1
        st, who = t(s1), t(t1)
        nat.check(nat.lib().gbl_greedy_act(st.data_ptr(), who.data_ptr(), None, h_dev.data_ptr(), depth, seed, base, call,
                                           out.data_ptr(), ch.data_ptr(), cm.data_ptr(), fb.data_ptr(), n, None))
        torch.cuda.synchronize()
        o = oracle.batch_greedy_act(s1, t1, h_ora, seed, base, call, depth=depth)
        for x, y in zip((out, ch, cm, fb), o):
            assert np.array_equal(npy(x), y), call
        assert np.array_equal(npy(h_dev), h_ora)
        fallbacks += int(o[3].sum())
        oracle.batch_step(s1, t1, d1, np.where(d1 != 0, 0, o[0]).astype(np.int32), auto_reset=True)
    assert fallbacks > 300
    # NULL optional outputs
    nat.check(nat.lib().gbl_greedy_act(st.data_ptr(), who.data_ptr(), None, h_dev.data_ptr(), 2, seed, base, 99,
                                       out.data_ptr(), None, None, None, n, None))
    torch.cuda.synchronize()
    o = oracle.batch_greedy_act(s1 * 0 + npy(st), npy(who), h_ora, seed, base, 99, depth=2)
    assert np.array_equal(npy(out), o[0]) and np.array_equal(npy(h_dev), h_ora)


def test_greedy_vs_oracle_selfplay(G):
    from gobblet_rl_amd import _native as nat
    state, tm, dn = selfplay(3000, 26, seed=8)
    live = oracle.batch_winner(state) == 0
    state, tm = np.ascontiguousarray(state[live]), np.ascontiguousarray(tm[live])
    n = len(state)
    rng = np.random.default_rng(2)
    hist = rng.integers(-1, 54, (n, 2, 3)).astype(np.int8)
    L = nat.lib()
    st, who, h = t(state), t(tm), t(hist)
    for depth in (1, 2):
        act = torch.empty(n, dtype=torch.int32, device=DEV); cm = torch.empty((n, 54), dtype=torch.int8, device=DEV)
        fb = torch.empty(n, dtype=torch.int8, device=DEV)
        nat.check(L.gbl_greedy(st.data_ptr(), who.data_ptr(), None, h.data_ptr(), depth, act.data_ptr(), cm.data_ptr(),
                               fb.data_ptr(), n, None))
        torch.cuda.synchronize()
        o = oracle.batch_greedy(state, tm, hist=hist, depth=depth)
        assert np.array_equal(npy(act), o[0]) and np.array_equal(npy(cm), o[1]) and np.array_equal(npy(fb), o[2])


# ---- full size (2^20 boards): one step against the oracle + invariants of a long rollout -----------------
def test_full_size_step_and_rollout_properties(G):
    n = 1 << 20
    env = G.BatchedGobblet(n, DEV, auto_reset=True, seed=0)
    env.rollout(64, count=True)           # warm-up plies of the benchmark workload
    state, tm = npy(env.squares), npy(env.to_move)
    # invariants of reachable states: every piece at most once, on its own level
    for k in range(3):
        lvl = state[:, 9 * k:9 * k + 9].astype(np.int16)
        assert np.isin(np.abs(lvl), [0, 2 * k + 1, 2 * k + 2]).all()
        for v in (2 * k + 1, 2 * k + 2, -(2 * k + 1), -(2 * k + 2)):
            assert ((lvl == v).sum(1) <= 1).all()
    assert (oracle.batch_winner(state) == 0).all()          # terminated boards were reset
    assert np.array_equal(npy(env.action_mask), oracle.batch_legal_mask(state, tm))
    c = npy(env.counters)
    assert c[0] == n * 64 and c[1] == c[2] + c[3] and 0.3 < c[2] / c[1] < 0.7
    # one more lockstep ply through sample + step, checked in full against the oracle
    a = npy(env.sample_actions())
    dn = np.zeros(n, np.int8)
    o = oracle.batch_step(state, tm, dn, a, auto_reset=True, threads=16)
    obs, rew, done, win = env.step(env.actions)
    assert np.array_equal(npy(env.squares), state) and np.array_equal(npy(done), dn)
    assert np.array_equal(npy(obs["action_mask"]), o["mask"]) and np.array_equal(npy(obs["observation"]), o["obs"])
    assert np.array_equal(npy(win), o["winner"])
    # determinism: the same seed replays the same trajectory; sharding does not change it
    e1 = G.BatchedGobblet(4096, DEV, auto_reset=True, seed=7, env_base=0)
    e2 = G.BatchedGobblet(2048, DEV, auto_reset=True, seed=7, env_base=2048)
    e1.rollout(50); e2.rollout(50)
    assert np.array_equal(npy(e1.squares)[2048:], npy(e2.squares))


def test_validate(G, golden_dir):
    g = np.load(os.path.join(golden_dir, "board_functions.npz"))
    sq = g["squares"][:100].copy()
    b = G.BatchedBoard(100, DEV, squares=t(sq))
    assert (npy(b.validate()) == 0).all()
    sq[7, 9:11] = (3, 3)
    b.squares = t(sq)
    with pytest.raises(Exception, match="PIECE HAS BEEN USED TWICE"):  # board.py:94-95
        b.validate()
    sq[7, 9:11] = (0, 0); sq[3, 0] = 5
    b.squares = t(sq)
    with pytest.raises(ValueError):
        b.validate()
    assert npy(b.validate(raise_on_error=False))[3] == 1


def test_error_behaviour(G):
    from gobblet_rl_amd import _native as nat
    L = nat.lib()
    x = torch.zeros(64 * 27 + 16, dtype=torch.int8, device=DEV)
    out = torch.zeros(64, dtype=torch.int8, device=DEV)
    assert L.gbl_winner(x.data_ptr() + 1, out.data_ptr(), 8, None) == nat.ERR_ALIGN
    assert b"aligned" in L.gbl_last_error()
    with pytest.raises(ValueError):
        G.BatchedGobblet(8, DEV).step(torch.zeros(9, dtype=torch.int32, device=DEV))
    with pytest.raises(ValueError):
        G.BatchedGobblet(8, DEV, illegal_mode="bogus")


# ---- the single-env AEC surface on the real engine (BASELINE.json configs[0] plumbing) -------------------
def test_aec_facade_on_gpu(G, golden_dir):
    g = np.load(os.path.join(golden_dir, "random_games.npz"))
    e = G.gobblet_v1.raw_env(device=DEV)
    for i in range(len(g["action"])):  # all 96 golden games, ply by ply, through the AEC surface
        if g["ply"][i] == 0:
            e.reset()
        e.step(int(g["action"][i]))
        assert np.array_equal(e.board.squares, g["squares_after"][i])
        o = e.observe(e.agent_selection)
        assert np.array_equal(o["action_mask"], g["mask_next"][i])
        if g["game"][i] < 24:
            assert np.array_equal(e.observe("player_1")["observation"], g["obs_p1"][i])
            assert np.array_equal(e.observe("player_2")["observation"], g["obs_p2"][i])
        assert [e.rewards["player_1"], e.rewards["player_2"]] == g["reward"][i].tolist()
        assert e.terminations["player_1"] == bool(g["done"][i])
    rng = np.random.default_rng(0)
    env = G.gobblet_v1.env(device=DEV)
    env.reset()
    totals = {"player_1": 0, "player_2": 0}
    for agent in env.agent_iter():  # examples/example_basic.py:50-67
        observation, reward, termination, truncation, info = env.last()
        totals[agent] += reward
        if termination or truncation:
            env.step(None)
        else:
            mask = observation["action_mask"]
            env.step(int(rng.choice(np.arange(len(mask)), p=mask / np.sum(mask))))
    assert sorted(totals.values()) == [-1, 1]


def test_greedy_policy_class(G, golden_dir):
    """Host mirror of GreedyGobbletPolicy: decisions from observations, history guard, fallback draw."""
    g = np.load(os.path.join(golden_dir, "greedy.npz"))
    pol = G.GreedyGobbletPolicy(depth=2, device=DEV)
    a = pol.compute_actions(g["obs"], g["mask"])
    ref = g["chosen_d2"].astype(np.int32)
    assert np.array_equal(npy(a), ref)               # empty history, a move was always chosen in this set
    assert np.array_equal(npy(pol.last_candidates), g["cands_d2"]) and not npy(pol.last_fallback).any()
    # second call on the same positions: the chosen action is now in the agent's last three -> fallback
    b = pol.compute_actions(g["obs"], g["mask"])
    assert npy(pol.last_fallback).all()
    picked = npy(b)
    assert (g["cands_d2"][np.arange(len(picked)), picked] == 1).all()  # drawn from actions_depth1
    # single-observation, reference-shaped call
    one = G.GreedyGobbletPolicy(depth=1, device=DEV).compute_action(g["obs"][320], g["mask"][320])
    assert int(one) == 8 and one.shape == ()
    assert G.gobblet_v1.GreedyGobbletPolicy is G.GreedyGobbletPolicy


def test_greedy_vs_greedy_games(G):
    """Two greedy agents play 2048 games in lockstep on the device (tutorials/GreedyAgent usage): runs to
    termination, and every move played was legal."""
    n = 2048
    env = G.BatchedGobblet(n, DEV, auto_reset=False)
    pol = G.GreedyGobbletPolicy(depth=2, device=DEV)
    for ply in range(2):  # first two plies random, as the tutorial does
        env.step(env.sample_actions())
    finished = 0
    for ply in range(60):
        live = npy(env.done) == 0
        if not live.any():
            break
        a = pol.compute_actions_from_state(env.squares, env.to_move)
        legal = npy(env.action_mask)[np.arange(n), np.clip(npy(a), 0, 53)]
        assert (legal[live] == 1).all()
        env.step(a)
    finished = int((npy(env.done) != 0).sum())
    assert finished > n * 0.9


def test_c1_thousand_reference_games(G, golden_dir):
    """BASELINE.md C1.  (a) all 12 009 plies of the 1000 reference games as one batched step;
    (b) the AEC loop of examples/example_basic.py:50-67 over gobblet_v1.env() with the same
    numpy.random.default_rng(0) stream: identical masks at every ply => identical action draws =>
    identical trajectories, game after game."""
    from tests.test_oracle_golden import c1_plies
    g, before, mover, mask = c1_plies(golden_dir)
    n = len(mover)
    env = vec_env(G, n, before, mover, np.zeros(n, np.int8))
    assert np.array_equal(npy(env.action_mask), mask)
    obs, rew, done, win = env.step(t(g["action"].astype(np.int32)))
    assert np.array_equal(npy(env.squares), g["squares_after"]) and np.array_equal(npy(win), g["winner"])
    rng = np.random.default_rng(0)
    e = G.gobblet_v1.env(device=DEV)
    i = 0
    for game in range(120):
        e.reset()
        for agent in e.agent_iter():
            observation, reward, termination, truncation, info = e.last()
            if termination or truncation:
                e.step(None)
                continue
            m = observation["action_mask"]
            a = int(rng.choice(np.arange(len(m)), p=m / np.sum(m)))
            assert a == g["action"][i]
            e.step(a)
            assert np.array_equal(e.unwrapped.board.squares, g["squares_after"][i])
            i += 1
        assert i == int(np.cumsum(g["game_len"])[game])


def test_checkpoint_resume(G, tmp_path):
    a = G.BatchedGobblet(5000, DEV, auto_reset=True, seed=3)
    a.rollout(20, count=True)
    torch.save(a.state_dict(), tmp_path / "env.pt")
    a.rollout(15, count=True)
    b = G.BatchedGobblet(5000, DEV, auto_reset=True)
    b.load_state_dict(torch.load(tmp_path / "env.pt"))
    b.rollout(15, count=True)
    assert torch.equal(a.squares, b.squares) and torch.equal(a.action_mask, b.action_mask)
    assert torch.equal(a.counters, b.counters) and a.ply == b.ply


def test_forty_million_boards_64bit_offsets(G):
    """40 M boards: the obs / mask buffers exceed 2^32 / 2^31 bytes, so every tile offset must be
    64-bit.  Slices around those byte boundaries and at both ends are checked against the oracle
    (boards are independent and the sampler is keyed by the global board id, so a slice can be
    replayed on its own)."""
    n, plies, seed = 40_000_000 + 13, 6, 5
    env = G.BatchedGobblet(n, DEV, auto_reset=True, seed=seed)
    env.rollout(plies)
    torch.cuda.synchronize()
    spots = [0, (1 << 31) // 117, (1 << 32) // 117, (1 << 31) // 54, (1 << 32) // 54, (1 << 30) // 27, n - 4096]
    for s0 in spots:
        s0 = max(0, min(n - 4096, s0 - 2048))
        st, tm, dn = oracle.batch_reset(4096)
        o = oracle.batch_rollout(st, tm, dn, seed, s0, 0, plies, threads=4)
        sl = slice(s0, s0 + 4096)
        assert np.array_equal(npy(env.squares[sl]), st), s0
        assert np.array_equal(npy(env.action_mask[sl]), o["mask"]), s0
        assert np.array_equal(npy(env.observation[sl]), o["obs"]), s0
        assert np.array_equal(npy(env.winner[sl]), o["winner"]) and np.array_equal(npy(env.actions[sl]), o["actions"])


def test_caller_shaped_adapters(G, golden_dir):
    """batches shaped like the reference's callers (Tianshou / RLlib adapters)."""
    g = np.load(os.path.join(golden_dir, "greedy.npz"))
    n = len(g["squares"])
    env = vec_env(G, n, g["squares"], g["to_move"], np.zeros(n, np.int8))
    tb = env.tianshou_batch()
    assert tb["obs"].shape == (n, 3, 3, 13) and tb["mask"].dtype == torch.bool and torch.equal(tb["agent_id"], env.to_move)
    out = G.GreedyGobbletPolicy(depth=2, device=DEV).forward({"obs": {"obs": tb["obs"], "mask": tb["mask"]}})
    assert np.array_equal(out["act"], g["chosen_d2"].astype(np.int64))
    rb = env.rllib_batch()
    assert rb["observation"].shape == (n, 117)
    acts = G.GreedyGobbletPolicy(depth=1, device=DEV).compute_actions_rllib({k: v.cpu().numpy() for k, v in rb.items()})
    ref = g["chosen_d1"].astype(np.int64)
    assert all(int(a) == r for a, r in zip(acts, ref) if r >= 0)
    rnd = G.RandomAdmissiblePolicy(seed=4, device=DEV)
    a = rnd.compute_actions(rb)
    assert np.array_equal(npy(a), oracle.batch_sample(g["mask"], 4, 0, 0))
    assert (g["mask"][np.arange(n), npy(a)] == 1).all()


def test_turn_counter(G, golden_dir):
    """raw_env.turn per board (gobblet.py:270,289): golden plies (turn = ply + 1 after every step, illegal
    no-ops included), then auto-reset steps and fused rollouts against the oracle."""
    g = np.load(os.path.join(golden_dir, "random_games.npz"))
    n = len(g["action"])
    env = G.BatchedGobblet(n, DEV, track_turn=True)
    env.board.squares = t(g["squares_before"]); env.to_move.copy_(t(g["mover"])); env.turn.copy_(t(g["ply"]))
    env.step(t(g["action"]))
    assert np.array_equal(npy(env.turn), g["ply"] + 1)
    n = 5000
    env = G.BatchedGobblet(n, DEV, auto_reset=True, seed=2, track_turn=True)
    s, tm, dn = oracle.batch_reset(n); turn = np.zeros(n, np.int32)
    for k in range(25):
        a = npy(env.sample_actions())
        env.step(env.actions)
        oracle.batch_step(s, tm, dn, a, auto_reset=True, turn=turn)
        assert np.array_equal(npy(env.turn), turn), k
    env.rollout(40)
    oracle.batch_rollout(s, tm, dn, 2, 0, 25, 40, turn=turn, threads=4)
    assert np.array_equal(npy(env.turn), turn) and np.array_equal(npy(env.squares), s) and turn.max() > 10
    before = npy(env.squares).copy()
    old = env.turn >= 8
    env.reset_where(old)  # truncation guard built from the turn counter
    sel = npy(old)
    assert (npy(env.squares)[sel] == 0).all() and np.array_equal(npy(env.squares)[~sel], before[~sel])
    assert (npy(env.turn)[sel] == 0).all() and (npy(env.action_mask)[sel] == 1).all()
    e2 = G.BatchedGobblet(64, DEV, illegal_mode="terminate", track_turn=True)
    e2.step(torch.full((64,), 60, dtype=torch.int32, device=DEV))  # illegal: the wrapper never calls raw step
    assert (npy(e2.turn) == 0).all() and (npy(e2.done) == 1).all()


@pytest.mark.parametrize("n", [1, 63, 64, 65, 4097])
def test_out_of_bounds_canaries(G, n):
    """Every output buffer sits between canary regions; no kernel may touch a byte outside
    [0, n * row) of any of them (ragged last tiles included)."""
    from gobblet_rl_amd import _native as nat
    L = nat.lib()
    PAD = 256  # bytes, multiple of 16 so the payload stays 16-byte aligned

    def guarded(nbytes):
        buf = torch.full((PAD + ((nbytes + 15) // 16) * 16 + PAD,), 0x5A, dtype=torch.uint8, device=DEV)
        return buf, buf[PAD:PAD + nbytes]

    def intact(buf, nbytes):
        b = buf.cpu().numpy()
        return (b[:PAD] == 0x5A).all() and (b[PAD + nbytes:] == 0x5A).all()

    state, tm, dn = selfplay(n, 12, seed=n)
    a = oracle.batch_sample(oracle.batch_legal_mask(state, tm), 1, 0, 0)
    sizes = {"state": 27 * n, "tm": n, "dn": n, "act": 4 * n, "win": n, "rew": 2 * n, "mask": 54 * n, "obs": 117 * n,
             "flat": 9 * n, "cov": 27 * n, "turn": 4 * n, "rec": 432 * n, "act2": 4 * n, "dn2": n, "tm2": n}
    bufs = {k: guarded(v) for k, v in sizes.items()}
    p = {k: bufs[k][1] for k in bufs}
    p["state"].copy_(t(state).view(torch.uint8).reshape(-1)); p["tm"].copy_(t(tm).view(torch.uint8))
    p["dn"].zero_(); p["turn"].zero_()
    p["act"].copy_(t(a).view(torch.uint8).reshape(-1))
    ptr = {k: v.data_ptr() for k, v in p.items()}
    for auto in (0, 1):
        nat.check(L.gbl_step(ptr["state"], ptr["tm"], ptr["dn"], ptr["act"], ptr["win"], ptr["rew"], ptr["mask"],
                             ptr["obs"], ptr["turn"], n, 0, auto, None))
        nat.check(L.gbl_rollout(ptr["state"], ptr["tm"], ptr["dn"], ptr["act"], ptr["win"], ptr["rew"], ptr["mask"],
                                ptr["obs"], n, 3, 0, 5, 2, 0, None, ptr["turn"], None))
        nat.check(L.gbl_step_into(ptr["state"], ptr["tm"], ptr["dn"], ptr["act"], ptr["win"], ptr["rew"], ptr["mask"],
                                  ptr["obs"], ptr["turn"], ptr["act2"], ptr["dn2"], ptr["tm2"], n, 0, auto, None))
    nat.check(L.gbl_legal_mask(ptr["state"], ptr["tm"], ptr["mask"], n, None))
    nat.check(L.gbl_observe(ptr["state"], ptr["tm"], -1, ptr["obs"], n, None))
    nat.check(L.gbl_sample(ptr["mask"], ptr["act"], n, 1, 0, 0, None))
    nat.check(L.gbl_winner(ptr["state"], ptr["win"], n, None))
    nat.check(L.gbl_flatboard(ptr["state"], ptr["flat"], n, None))
    nat.check(L.gbl_covered(ptr["state"], ptr["cov"], n, None))
    nat.check(L.gbl_is_legal(ptr["state"], ptr["tm"], ptr["act"], ptr["win"], n, None))
    nat.check(L.gbl_play_turn(ptr["state"], ptr["tm"], ptr["act"], n, None))
    nat.check(L.gbl_validate(ptr["state"], ptr["win"], n, None))
    nat.check(L.gbl_decode_obs(ptr["obs"], ptr["cov"], ptr["win"], n, None))
    nat.check(L.gbl_greedy(ptr["state"], ptr["tm"], None, None, 2, ptr["act"], ptr["mask"], ptr["win"], n, None))
    nat.check(L.gbl_board_eval(ptr["state"], ptr["tm"], ptr["act"], ptr["rec"], n, None))
    nat.check(L.gbl_board_eval(ptr["state"], None, None, ptr["rec"], n, None))
    torch.cuda.synchronize()
    for k, (buf, _) in bufs.items():
        assert intact(buf, sizes[k]), k
    # the trajectory entry points (gbl_collect, gbl_collect_from, gbl_collect_policy) and gbl_greedy_act: arrays of exactly
    # the cells the ABI addresses -- (T - 1) * ply_stride + n -- with the smallest legal slot (n rounded up to whole tiles)
    T, slot = 3, -(-n // 64) * 64
    cells = (T - 1) * slot + n
    tsizes = {"act": 4 * cells, "win": cells, "rew": 2 * cells, "dn": cells, "tm": cells, "mask": 54 * cells, "obs": 117 * cells,
              "chosen": 4 * cells, "how": cells, "cand": 54 * cells, "hist": 6 * n, "fin": 4 * n, "fb": n, "cm": 54 * n}
    tb = {k: guarded(v) for k, v in tsizes.items()}
    q = {k: tb[k][1].data_ptr() for k in tb}
    tb["hist"][1].fill_(255)
    for first in (None, ptr["act"]):
        nat.check(L.gbl_collect_from(ptr["state"], ptr["tm"], ptr["dn"], first, q["act"], q["win"], q["rew"], q["dn"], q["tm"], q["mask"],
                                     q["obs"], n, slot, 64, 7, 0, 11, None, T, 0, None, ptr["turn"], None))
    for pol in ((2, 2), (1, 0), (0, 3)):
        nat.check(L.gbl_collect_policy(ptr["state"], ptr["tm"], ptr["dn"], q["hist"], q["act"], q["win"], q["rew"], q["dn"], q["tm"],
                                       q["mask"], q["obs"], q["chosen"], q["how"], q["cand"], n, slot, 64, 7, 0, 20, None, T, pol[0],
                                       pol[1], 1, 0, None, ptr["turn"], None))
    nat.check(L.gbl_greedy_act(ptr["state"], ptr["tm"], None, q["hist"], 2, 7, 0, 3, q["fin"], ptr["act"], q["cm"], q["fb"], n, None))
    torch.cuda.synchronize()
    for k, (buf, _) in tb.items():
        assert intact(buf, tsizes[k]), k
    for k in ("state", "tm", "dn", "turn", "act"):
        assert intact(bufs[k][0], sizes[k]), k


def test_c_abi_without_python(G, tmp_path):
    """The boundary is a plain C-ABI: examples/c_abi_example.cpp drives it with hipMalloc'ed buffers only."""
    import shutil
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    exe = str(tmp_path / "c_abi_example")
    csrc = os.path.join(root, "gobblet-rl_amd", "csrc")
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-O2", "-I", os.path.join(root, "include"),
                           os.path.join(root, "examples", "c_abi_example.cpp"), "-L", csrc, "-lgobblet_hip",
                           f"-Wl,-rpath,{csrc}", "-o", exe])
    out = subprocess.run([exe, "131072", "30"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "misaligned pointer refused" in out.stdout
    # the example's own copy of the placement search (gbl_block_alloc / gbl_placement_probe / gbl_block_free) found a pair
    assert "trajectory arrays placed: probe ratio" in out.stdout and "kernel variant 2" in out.stdout


def test_example_scripts_run(G):
    """examples/ run as a user would run them (child processes): the reference's example_basic loop over
    the AEC surface, and the batched throughput loop with both policies."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    runs = [["examples/example_basic.py", "--seed", "3", "--render_mode", "text"],
            ["examples/example_batched.py", "--boards", "65536", "--plies", "40", "--policy", "random"],
            ["examples/example_batched.py", "--boards", "4096", "--plies", "12", "--policy", "greedy"],
            ["examples/example_batched.py", "--boards", "65536", "--plies", "40", "--policy", "random", "--graph", "8"],
            ["examples/example_batched.py", "--boards", "4096", "--plies", "12", "--policy", "greedy", "--graph", "4"],
            ["examples/example_batched.py", "--boards", "65536", "--plies", "48", "--policy", "random", "--collect", "16"],
            ["examples/example_batched.py", "--boards", "16384", "--plies", "32", "--policy", "greedy", "--opponent", "random",
             "--collect", "16"],
            # an external policy against the masked-random draw the step itself leaves behind (gbl_step_ex)
            ["examples/example_batched.py", "--boards", "8192", "--plies", "24", "--policy", "greedy", "--opponent", "random"]]
    for cmd in runs:
        r = subprocess.run([sys.executable] + cmd, cwd=root, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        if "basic" in cmd[0]:
            assert r.stdout.count("Reward:") == 2 and "Reward: 1" in r.stdout and "Reward: -1" in r.stdout
        else:
            assert "env-steps/s" in r.stdout and "games finished" in r.stdout


def test_graph_replay_draws_fresh_plies(G):
    """A captured hipGraph of K rollout plies + advance_ply(), replayed R times, plays K*R DIFFERENT plies: the
    ply index lives on the device (gbl_rollout_at / gbl_counter_add).  Compared with the oracle's K*R plies; the
    same for sample + step, and for greedy policy steps (fallback draws keyed by a device-resident call index)."""
    n, K, R, seed, base = 3000, 3, 4, 11, 500
    env = G.BatchedGobblet(n, DEV, auto_reset=True, seed=seed, env_base=base)
    env.rollout(2)                       # some history first: the device counter starts at 2
    env.device_ply()
    s_ = torch.cuda.Stream()
    s_.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s_):
        env.rollout(1); env.advance_ply()   # warm-up on the side stream (ply 2)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, capture_error_mode="thread_local"):
            for _ in range(K):
                env.rollout(1)
            env.advance_ply()
        for _ in range(R):
            g.replay()
    torch.cuda.current_stream().wait_stream(s_)
    torch.cuda.synchronize()
    s, tm, dn = oracle.batch_reset(n)
    o = oracle.batch_rollout(s, tm, dn, seed, base, 0, 3 + K * R)
    assert env.ply == 3 + K * R
    assert np.array_equal(npy(env.squares), s) and np.array_equal(npy(env.action_mask), o["mask"])
    assert np.array_equal(npy(env.actions), o["actions"]) and np.array_equal(npy(env.observation), o["obs"])
    # eager calls in device mode, then back to the host counter: the stream of draws just continues
    env.step(env.sample_actions())
    env.device_ply(False)
    env.rollout(2)
    a = oracle.batch_sample(oracle.batch_legal_mask(s, tm), seed, base, 3 + K * R)
    oracle.batch_step(s, tm, dn, a, auto_reset=True)
    oracle.batch_rollout(s, tm, dn, seed, base, 4 + K * R, 2)
    assert env.ply == 6 + K * R and np.array_equal(npy(env.squares), s)
    # greedy policy steps: the call index on the device
    pol = G.GreedyGobbletPolicy(depth=2, seed=3, device=DEV)
    pol.device_calls()
    st, who = env.squares.clone(), env.to_move.clone()
    outs = []
    for _ in range(3):
        outs.append(npy(pol.compute_actions_from_state(st, who)).copy())
        pol.advance_calls()
    hist = np.full((n, 2, 3), -1, np.int8)
    for call in range(3):
        og = oracle.batch_greedy_act(npy(st), npy(who), hist, 3, 0, call, depth=2)
        assert np.array_equal(outs[call], og[0]), call
    assert np.array_equal(npy(pol.prev_actions), hist)


# ---- at-size checks (BASELINE configs at their stated sizes) -------------------------------------------
def test_greedy_config5_full_size(G):
    """BASELINE config 5 as stated: 65 536 boards of the stationary masked-random mix (both movers) x depth-2
    greedy, one launch = 1 024 tiles x 4 wavefronts with pooled LDS candidate lists; every decision, candidate
    set and fallback flag against the oracle -- with empty histories and with random ones."""
    from gobblet_rl_amd import _native as nat
    n = 65536
    env = G.BatchedGobblet(n, DEV, auto_reset=True, seed=11)
    env.rollout(64)
    torch.cuda.synchronize()
    state, tm = npy(env.squares), npy(env.to_move)
    assert (oracle.batch_winner(state) == 0).all() and 0.3 < tm.mean() < 0.7
    rng = np.random.default_rng(5)
    hist = rng.integers(-1, 54, (n, 2, 3)).astype(np.int8)
    L = nat.lib()
    act = torch.empty(n, dtype=torch.int32, device=DEV); cm = torch.empty((n, 54), dtype=torch.int8, device=DEV)
    fb = torch.empty(n, dtype=torch.int8, device=DEV)
    for h in (None, hist):
        hd = None if h is None else t(h)
        nat.check(L.gbl_greedy(env.squares.data_ptr(), env.to_move.data_ptr(), None, nat.ptr(hd), 2, act.data_ptr(),
                               cm.data_ptr(), fb.data_ptr(), n, nat.current_stream(torch.device(DEV))), "gbl_greedy")
        torch.cuda.synchronize()
        o = oracle.batch_greedy(state, tm, hist=h, depth=2, threads=16)
        assert np.array_equal(npy(act), o[0]) and np.array_equal(npy(cm), o[1]) and np.array_equal(npy(fb), o[2])
        assert h is None or 0 < int(o[2].sum()) < n  # with histories both outcomes of the fallback test occur


@pytest.mark.parametrize("n", [20000, 98304, 300000])
def test_greedy_every_launch_shape(G, n):
    """gbl_greedy picks its workgroup shape by batch size (greedy_shape): <1,16> up to 16 384 boards (the golden and
    self-play tests), <2,8> up to 32 768, <4,16> for whole generations of 65 536 boards up to 262 144 (config 5's test),
    <1,4> in between and beyond.  The other three, each against the oracle on the masked-random mix with random histories
    (a ragged last tile included)."""
    from gobblet_rl_amd import _native as nat
    n = n + 37
    env = G.BatchedGobblet(n, DEV, auto_reset=True, seed=n)
    env.rollout(48)
    torch.cuda.synchronize()
    state, tm = npy(env.squares), npy(env.to_move)
    hist = np.random.default_rng(n).integers(-1, 54, (n, 2, 3)).astype(np.int8)
    act = torch.empty(n, dtype=torch.int32, device=DEV); cm = torch.empty((n, 54), dtype=torch.int8, device=DEV)
    fb = torch.empty(n, dtype=torch.int8, device=DEV)
    nat.check(nat.lib().gbl_greedy(env.squares.data_ptr(), env.to_move.data_ptr(), None, t(hist).data_ptr(), 2, act.data_ptr(),
                                   cm.data_ptr(), fb.data_ptr(), n, nat.current_stream(torch.device(DEV))), "gbl_greedy")
    torch.cuda.synchronize()
    o = oracle.batch_greedy(state, tm, hist=hist, depth=2, threads=16)
    assert np.array_equal(npy(act), o[0]) and np.array_equal(npy(cm), o[1]) and np.array_equal(npy(fb), o[2])
    assert 0 < int(o[2].sum()) < n


def test_greedy_on_terminal_roots(G, golden_dir):
    """gbl_greedy on roots that already hold a line (the wavefront-uniform fallback of the root analysis to the general form of
    outcomes54), mixed with live positions so that wavefronts of both kinds occur, against the oracle."""
    from gobblet_rl_amd import _native as nat
    from tests.positions import terminal_roots
    term = terminal_roots(np.load(os.path.join(golden_dir, "board_functions.npz")))
    env = G.BatchedGobblet(8192, DEV, auto_reset=True, seed=2)
    env.rollout(40)
    torch.cuda.synchronize()
    state = np.ascontiguousarray(np.concatenate([npy(env.squares)[:4096], term, npy(env.squares)[4096:]]))
    n = len(state)
    tm = (np.arange(n) % 2).astype(np.int8)
    act = torch.empty(n, dtype=torch.int32, device=DEV); cm = torch.empty((n, 54), dtype=torch.int8, device=DEV)
    fb = torch.empty(n, dtype=torch.int8, device=DEV)
    st_d, tm_d = t(state), t(tm)
    for depth in (1, 2):
        nat.check(nat.lib().gbl_greedy(st_d.data_ptr(), tm_d.data_ptr(), None, None, depth, act.data_ptr(), cm.data_ptr(),
                                       fb.data_ptr(), n, nat.current_stream(torch.device(DEV))), "gbl_greedy")
        torch.cuda.synchronize()
        o = oracle.batch_greedy(state, tm, depth=depth, threads=16)
        assert np.array_equal(npy(act), o[0]) and np.array_equal(npy(cm), o[1]) and np.array_equal(npy(fb), o[2]), depth


def test_winner_exhaustive_on_gpu(G):
    """gbl_winner over every pattern of tops (3^9, at each level) and 100 000 random stacks (the packed line
    arithmetic of winner_of against the oracle's walk over the 8 lines), and the same boards through the fused
    step's winner output (an illegal action 54 leaves the board as it is, raw_env semantics)."""
    import itertools
    cfg = np.array(list(itertools.product((0, 1, -1), repeat=9)), np.int8)
    parts = []
    for lvl in range(3):
        st = np.zeros((len(cfg), 27), np.int8)
        st[:, 9 * lvl:9 * lvl + 9] = cfg * (2 * lvl + 1)
        parts.append(st)
    rng = np.random.default_rng(1)
    st = np.zeros((100000, 27), np.int8)
    for lvl, vals in enumerate(((1, 2), (3, 4), (5, 6))):
        occ = rng.random((len(st), 9)) < 0.45
        v = rng.choice(vals, size=(len(st), 9)) * rng.choice((1, -1), size=(len(st), 9))
        st[:, 9 * lvl:9 * lvl + 9] = np.where(occ, v, 0)
    parts.append(st)
    boards = np.ascontiguousarray(np.concatenate(parts))
    w = oracle.batch_winner(boards)
    assert len(np.unique(w)) == 3
    b = G.BatchedBoard(len(boards), DEV, squares=t(boards))
    assert np.array_equal(npy(b.check_for_winner()), w)
    env = vec_env(G, len(boards), boards, np.zeros(len(boards), np.int8), np.zeros(len(boards), np.int8))
    env.step(torch.full((len(boards),), 54, dtype=torch.int32, device=DEV))
    assert np.array_equal(npy(env.winner), w) and np.array_equal(npy(env.squares), boards)


def _norm_np_repr(text):
    import re
    return re.sub(r"np\.int64\((-?\d+)\)", r"\1", text)


def test_text_render_and_debug_printers_on_gpu(G, golden_dir, capsys):
    """SURVEY 8 f3 on the HIP engine: every frame of render_mode "text" / "text_full" (gobblet.py:299-429) along
    the golden games, Board.print_pieces / print / __str__ (board.py:155-156, 223-242) and the args.debug frames
    (gobblet.py:315-317), character for character against stdout captured from the reference."""
    import types
    frames = json.load(open(os.path.join(golden_dir, "render_text.json")))
    g = np.load(os.path.join(golden_dir, "random_games.npz"))
    for mode in ("text", "text_full"):
        e = G.gobblet_v1.raw_env(render_mode=mode, device=DEV)
        for fr in [f for f in frames if f["mode"] == mode]:
            i = fr["index"]
            if g["ply"][i] == 0:
                e.reset()
            capsys.readouterr()
            e.step(int(g["action"][i]))
            assert capsys.readouterr().out == fr["text"], (mode, i)
    ref = json.load(open(os.path.join(golden_dir, "print_pieces.json")))
    for rec in ref["boards"]:
        b = G.gobblet_v1.Board(squares=rec["squares"], device=DEV)
        for key, fn in (("print_pieces", b.print_pieces), ("print", b.print), ("str", lambda: print(str(b)))):
            capsys.readouterr()
            fn()
            assert _norm_np_repr(capsys.readouterr().out) == _norm_np_repr(rec[key]), (rec["index"], key)
    e = G.gobblet_v1.raw_env(render_mode="text", args=types.SimpleNamespace(debug=True), device=DEV)
    for fr in ref["debug_frames"]:
        i = fr["index"]
        if g["ply"][i] == 0:
            e.reset()
        capsys.readouterr()
        e.step(int(g["action"][i]))
        assert _norm_np_repr(capsys.readouterr().out) == _norm_np_repr(fr["text"]), i


def test_debug_branch_with_illegal_plies_on_gpu(G, golden_dir, capsys):
    """raw_env.step's `--ERROR-- ILLEGAL MOVE` branch (gobblet.py:238-242, agent-name quirk included) on the GPU-backed
    facade against the reference's captured stdout: the same check the CPU suite runs on the stand-in engine."""
    from tests.test_abi_and_host import _check_debug_illegal_frames
    _check_debug_illegal_frames(G, capsys, golden_dir, device=DEV)


def test_load_state_dict_twice_and_graph_after_load(G):
    """load_state_dict copies into the environment's own tensors: the checkpoint is not aliased (loading it again
    rewinds to the same state), tensor addresses survive (a hipGraph captured before the load keeps working), and
    `turn` travels with it."""
    n = 3000
    env = G.BatchedGobblet(n, DEV, auto_reset=True, seed=4, track_turn=True)
    env.rollout(17)
    sd = env.state_dict()
    keep = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in sd.items()}
    ptrs = (env.squares.data_ptr(), env.to_move.data_ptr(), env.action_mask.data_ptr())
    env.rollout(9)
    after9 = (env.squares.clone(), env.turn.clone(), env.action_mask.clone())
    for _ in range(2):  # rewind twice from the SAME dict
        env.load_state_dict(sd)
        assert ptrs == (env.squares.data_ptr(), env.to_move.data_ptr(), env.action_mask.data_ptr())
        assert all(torch.equal(sd[k], keep[k]) for k in sd if torch.is_tensor(sd[k]))  # the checkpoint is untouched
        assert torch.equal(env.turn, keep["turn"]) and env.ply == keep["ply"]
        env.rollout(9)
        assert all(torch.equal(a, b) for a, b in zip(after9, (env.squares, env.turn, env.action_mask)))
    # a graph captured before a load plays on the restored state
    env.load_state_dict(sd)
    env.device_ply()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        env.rollout(1); env.advance_ply()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, capture_error_mode="thread_local"):
            for _ in range(4):
                env.rollout(1)
            env.advance_ply()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    env.load_state_dict(sd)
    torch.cuda.synchronize()
    with torch.cuda.stream(side):
        g.replay(); g.replay()
    torch.cuda.synchronize()
    ref = G.BatchedGobblet(n, DEV, auto_reset=True, seed=4, track_turn=True)
    ref.load_state_dict(sd)
    ref.rollout(8)
    assert torch.equal(env.squares, ref.squares) and torch.equal(env.turn, ref.turn) and env.ply == ref.ply
    with pytest.raises(ValueError):
        G.BatchedGobblet(8, DEV, auto_reset=False).rollout(1)


@pytest.mark.parametrize("n,with_obs,illegal", [(65, True, "noop"), (4099, True, "noop"), (4096, False, "terminate"),
                                                 (1, True, "noop"),
                                                 # 2560 tiles: the largest grid of the two-wavefronts-per-tile kernel
                                                 # (k_collect2); one board more: the first of k_collect
                                                 (163840, True, "noop"), (163841, True, "terminate")])
def test_collect_equals_ply_by_ply_rollout(G, n, with_obs, illegal):
    """gbl_collect: T plies in one launch, every ply materialised in its trajectory slot == T launches of the
    fused single ply (gbl_rollout, plies = 1), which is itself checked against the oracle; state, turn and tallies
    after the launch included.  Ragged batch sizes exercise the slot stride (slot_boards = n rounded up to 16)."""
    T, seed, base = 23, 6, 10_000_000_000
    kw = dict(auto_reset=True, seed=seed, env_base=base, with_observation=with_obs, illegal_mode=illegal, track_turn=True)
    a, b = G.BatchedGobblet(n, DEV, **kw), G.BatchedGobblet(n, DEV, **kw)
    a.rollout(5, count=True); b.rollout(5, count=True)  # not from the empty board, not from ply 0
    tr = a.collect(T, count=True)
    assert tr["_slot_boards"] % 16 == 0 and tr["_slot_boards"] >= n
    a2 = G.BatchedGobblet(n, DEV, **kw)  # the same plies into tile-major buffers: (tiles, plies, 64, ...)
    a2.rollout(5, count=True)
    tt = a2.collect(T, count=True, layout="tile")
    bb, ll = torch.arange(n, device=DEV) // 64, torch.arange(n, device=DEV) % 64
    for key in ("actions", "winner", "rewards", "done", "to_move", "action_mask") + (("observation",) if with_obs else ()):
        assert tt[key].shape[:3] == (-(-n // 64), T, 64)
        assert torch.equal(tt[key][bb, :, ll].transpose(0, 1), tr[key]), key
    assert torch.equal(a2.squares, a.squares) and torch.equal(a2.action_mask, a.action_mask) and torch.equal(a2.turn, a.turn)
    for t_ in range(T):
        b.rollout(1, count=True)
        assert torch.equal(tr["actions"][t_], b.actions), t_
        assert torch.equal(tr["winner"][t_], b.winner) and torch.equal(tr["rewards"][t_], b.rewards), t_
        assert torch.equal(tr["done"][t_], b.done) and torch.equal(tr["to_move"][t_], b.to_move), t_
        assert torch.equal(tr["action_mask"][t_], b.action_mask), t_
        if with_obs:
            assert torch.equal(tr["observation"][t_], b.observation), t_
    assert torch.equal(a.squares, b.squares) and torch.equal(a.to_move, b.to_move) and torch.equal(a.done, b.done)
    assert torch.equal(a.turn, b.turn) and torch.equal(a.counters, b.counters) and a.ply == b.ply
    assert torch.equal(a.action_mask, b.action_mask)
    assert int(tr["done"].sum()) > 0 or n == 1  # games ended (and restarted) inside the trajectory
    # the padding boards of a slot are never written
    pad = tr["_full"]["action_mask"][:, n:]
    assert pad.numel() == 0 or int(pad.abs().sum()) == 0


def test_collect_vs_oracle_and_graph_replay(G):
    """A trajectory against the oracle ply by ply, then the same launch replayed from a hipGraph with the ply
    index on the device (fresh plies on every replay)."""
    n, T, seed = 3000, 12, 21
    env = G.BatchedGobblet(n, DEV, auto_reset=True, seed=seed)
    s, tm, dn = oracle.batch_reset(n)
    tr = env.collect(T)
    torch.cuda.synchronize()
    for t_ in range(T):
        o = oracle.batch_rollout(s, tm, dn, seed, 0, t_, 1)
        assert np.array_equal(npy(tr["actions"][t_]), o["actions"]) and np.array_equal(npy(tr["action_mask"][t_]), o["mask"])
        assert np.array_equal(npy(tr["observation"][t_]), o["obs"]) and np.array_equal(npy(tr["to_move"][t_]), tm)
    assert np.array_equal(npy(env.squares), s)
    env.device_ply()
    buf = env.trajectory_buffers(T)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        env.collect(T, out=buf); env.advance_ply()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, capture_error_mode="thread_local"):
            env.collect(T, out=buf)
            env.advance_ply()
        g.replay(); g.replay()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    oracle.batch_rollout(s, tm, dn, seed, 0, T, 2 * T)           # eager launch + first replay
    last = [oracle.batch_rollout(s, tm, dn, seed, 0, 3 * T + t_, 1) for t_ in range(T)]  # second replay, ply by ply
    assert env.ply == 4 * T and np.array_equal(npy(env.squares), s)
    assert all(np.array_equal(npy(buf["observation"][t_]), last[t_]["obs"]) for t_ in range(T))


def _bench(args, env=None, timeout=600, launcher_ranks=0):
    """Run bench.py (optionally under torch.distributed.run) with the full record sent to a temporary file; returns
    (stdout lines that are JSON, the full record, stderr)."""
    import socket
    import subprocess
    import sys
    import tempfile
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with tempfile.TemporaryDirectory() as tmp:
        full_path = os.path.join(tmp, "full.json")
        cmd = [os.path.join(root, "bench.py"), *args, "--configs-out", full_path]
        if launcher_ranks:
            with socket.socket() as sock:  # a free port for the rendezvous
                sock.bind(("127.0.0.1", 0))
                port = sock.getsockname()[1]
            cmd = ["-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(launcher_ranks), "--master-addr",
                   "127.0.0.1", "--master-port", str(port)] + cmd
        out = subprocess.run([sys.executable] + cmd, capture_output=True, text=True, timeout=timeout, cwd=root,
                             env=env if env is not None else dict(os.environ, MASTER_ADDR="127.0.0.1"))
        assert out.returncode == 0, out.stderr[-3000:]
        full = json.load(open(full_path))
    return [ln for ln in out.stdout.splitlines() if ln.startswith("{")], full, out


CONTRACT_KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                 "vs_baseline", "dtype", "data", "config", "roofline")


def test_bench_script_runs_and_reports(G):
    """bench.py's contract line on a small shard, in each mode: one JSON line with the contract's keys, a roofline
    whose achieved rate is consistent with the reported time, and the sub-records it promises."""
    for extra in (["--mode", "collect"], ["--mode", "fused"], ["--mode", "step"], ["--mode", "collect", "--no-obs", "--graph", "0"]):
        lines, full, out = _bench(["--boards", "16384", "--steps", "20", "--warmup", "5", "--no-configs", "--no-cpu-baseline", *extra])
        assert len(lines) == 1 and out.stdout.rstrip().splitlines()[-1] == lines[0]
        d = json.loads(lines[0])
        for k in CONTRACT_KEYS:
            assert k in d, k
        assert d["steps"] == 20 and d["n_gpus"] == 1 and d["config"]["total_boards"] == 16384 and d["scaling"] == "strong"
        assert all(not isinstance(v, (list, dict)) for v in d["config"].values())     # a workload string + scalars
        r = d["roofline"]
        # (the compact line carries six significant digits of everything but `value` / `ms_per_step`)
        assert r["bound"] == "hbm" and 0 < r["frac"] < 1 and abs(r["achieved"] / r["peak"] - r["frac"]) < 1e-5
        assert abs(d["value"] - 16384 * 20 / (d["ms_per_step"] * 20 / 1e3)) / d["value"] < 1e-6
        assert full["value"] == d["value"] and abs(full["roofline"]["frac"] - r["frac"]) < 1e-5
        assert r["algorithmic_bytes_survey"] in (234, 117) and r["frac_on_survey_bytes"] >= r["frac"] - 1e-5 and r["accounting"]
    # the RCCL path of an N > 1 run -- process group on the device, barriers around the timed region, MAX-reduce of the elapsed
    # time, all-gather of the per-rank numbers -- rehearsed at world size 1 (an 8-GPU node is the driver's to launch)
    env = dict(os.environ, GBL_BENCH_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29511", RANK="0", WORLD_SIZE="1",
               LOCAL_RANK="0")
    lines, full, out = _bench(["--boards", "262144", "--steps", "20", "--warmup", "5", "--no-configs", "--no-cpu-baseline"], env=env)
    # (RCCL's version banner and anything else a library prints goes to stderr: ONE line on stdout)
    assert len(out.stdout.strip().splitlines()) == 1, out.stdout[:600]
    d = json.loads(lines[-1])
    assert d["config"]["rccl_ranks"] == 1 and d["config"]["dist_backend"] == "nccl" and len(full["detail"]["kernel_us_per_rank"]) == 1
    assert d["config"]["span_barrier"] == "node-local shared memory"
    assert full["detail"]["trajectory_placement_per_rank"][0]["probes"]
    assert abs(d["config"]["kernel_us_max"] / full["detail"]["kernel_us_per_rank"][0] - 1) < 1e-5
    assert d["ms_per_step"] >= d["config"]["ms_per_step_own_span"] * (1 - 1e-5) > 0


def test_bench_driver_command_prints_a_compact_line(G):
    """The driver's EXACT command -- python3 bench.py --gpus 1 --steps 20 --warmup 5, sub-records and CPU baseline ON -- must
    end stdout with ONE compact line (< 4 096 bytes: the driver keeps only the tail of stdout; round 3's 20 kB line came back
    `parsed: null`) that carries the contract's keys, a flat roofline with `frac` and `traffic`, and a flat cpu_baseline; the 18
    sub-records are in the file config.configs_file names.  And: the same command under a launcher with WORLD_SIZE=1 (how a
    SCALE run's N = 1 leg is started) reports the same workload and kernel."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench
    out = subprocess.run([sys.executable, "bench.py", "--gpus", "1", "--steps", "20", "--warmup", "5"], capture_output=True,
                         text=True, timeout=900, cwd=root)
    assert out.returncode == 0, out.stderr[-3000:]
    last = out.stdout.rstrip("\n").splitlines()[-1]
    assert len(last.encode()) < 4096, len(last)
    d = json.loads(last)
    for k in CONTRACT_KEYS + ("cpu_baseline",):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 20 and d["warmup"] == 5 and d["config"]["total_boards"] == 1 << 20
    assert all(not isinstance(v, (list, dict)) for sub in ("config", "roofline", "cpu_baseline") for v in d[sub].values())
    r, c = d["roofline"], d["cpu_baseline"]
    assert r["bound"] == "hbm" and 0.3 < r["frac"] < 1 and "traffic" in r and "traffic_over_algorithmic" in r
    if r["traffic"] is not None:   # (null only while the committed counters are older than the kernel sources)
        assert 0.9 < r["traffic_over_algorithmic"] < 1.5
    assert c["value"] > 0 and c["cores"] >= 1 and c["kind"] == "port" and c["unit"] == "env-steps/s" and c["value_1core"] > 0
    assert abs(d["value"] - (1 << 20) * 20 / (d["ms_per_step"] * 20 / 1e3)) / d["value"] < 1e-6
    full = json.load(open(os.path.join(root, d["config"]["configs_file"])))
    assert set(full["configs"]) == set(bench.CONFIG_RECORDS) | set(bench.EXTRA_RECORDS)
    assert d["config"]["configs_recorded"] == len(full["configs"]) and full["value"] == d["value"]
    # every BASELINE config, both byte accountings and the predicted scaling curve ride in the parsed line
    cc = d["configs_compact"]
    for name in ("headline_recipe_1048576", "c2_4096", "c3_262144", "c4_shard_131072", "c5_greedy_65536", "single_ply_1048576",
                 "step_pipeline_1048576"):
        v, us, frac = cc[name]
        assert v > 1e9 and us > 0 and (frac is None or 0.02 < frac < 1.0), (name, cc[name])
        assert abs(v / full["configs"][name]["value"] - 1) < 1e-3
    assert r["algorithmic_bytes_survey"] == 234 and r["frac_on_survey_bytes"] > r["frac"] and len(r["accounting"]) <= 128
    sp = d["scale_prediction"]
    assert set(sp) == {"2", "4", "8"} and all(0.3 < sp[k][1] < 1.15 for k in sp) and sp["8"][0] > d["value"]
    assert all("roofline" in rec and rec["value"] > 0 for rec in full["configs"].values())
    # N = 1 under a launcher: same workload string, same kernel
    lines, _, _ = _bench(["--gpus", "1", "--steps", "20", "--warmup", "5", "--no-configs", "--no-cpu-baseline"], launcher_ranks=1)
    e = json.loads(lines[-1])
    assert e["config"]["workload"] == d["config"]["workload"] and e["roofline"]["kernel"] == d["roofline"]["kernel"]
    assert e["config"]["plies_per_launch"] == d["config"]["plies_per_launch"] and e["n_gpus"] == 1
    assert e["config"]["rccl_ranks"] == 0 and e["config"]["dist_backend"] is None  # (world 1: no process group, as in the bare form)


def test_bench_script_two_ranks_rehearsal(G):
    """The N > 1 path of bench.py (one rank per GPU under torch.distributed.run: barriers around the timed region,
    MAX-reduce of the elapsed time, all-gather of the per-rank kernel times, strong-scaling shards) rehearsed with two
    ranks that share this box's one GPU over gloo."""
    args = ["--gpus", "2", "--steps", "20", "--warmup", "5", "--boards", "32768", "--dist-backend", "gloo", "--share-device"]
    clean = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    # under the launcher (the driver's form), and bare: bench.py then starts its two ranks itself
    for ranks, environ in ((2, None), (0, clean)):
        lines, full, _ = _bench(args, env=environ, launcher_ranks=ranks)
        assert len(lines) == 1  # rank 0 only
        d = json.loads(lines[0])
        assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["config"]["total_boards"] == 32768
        assert d["config"]["boards_per_gpu"] == 16384 and len(full["detail"]["kernel_us_per_rank"]) == 2
        assert len(full["detail"]["trajectory_placement_per_rank"]) == 2 and d["config"]["dist_backend"] == "gloo"
        assert d["config"]["rccl_ranks"] == 0                   # (gloo rehearsal; under RCCL this is the world size)
        assert abs(d["config"]["kernel_us_max"] / max(full["detail"]["kernel_us_per_rank"]) - 1) < 1e-5
        assert "configs" not in full and "cpu_baseline" not in d  # N = 1 only


def test_bench_script_c4_shape_rehearsal(G):
    """BASELINE C4's shape -- 131 072 boards per rank (k_collect2, 20 plies in one launch) -- with as many ranks as one
    box allows on its card: FOUR ranks sharing the GPU over gloo (gpurun's process guard allows six processes on the card
    and this test process holds it too; the eight-rank run is the driver's).  Catches what one or two ranks do not: the
    N-way lock around the library build / load, the rendezvous, N placement searches on one device (each capped by what
    is free: they must fall back quietly), the all-gather of per-rank numbers."""
    ranks = 4
    lines, full, out = _bench(["--gpus", str(ranks), "--steps", "20", "--warmup", "5", "--boards", str(131072 * ranks),
                               "--dist-backend", "gloo", "--share-device"], launcher_ranks=ranks, timeout=900)
    assert len(lines) == 1 and len(lines[0].encode()) < 4096
    assert len(out.stdout.strip().splitlines()) == 1, out.stdout[:600]  # the launcher's stdout: rank 0's contract line, nothing else
    d = json.loads(lines[0])
    assert d["n_gpus"] == ranks and d["config"]["boards_per_gpu"] == 131072 and d["config"]["plies_per_launch"] == 20
    assert d["config"]["span_barrier"] == "node-local shared memory" and d["config"]["launch"] == "eager launches"
    assert d["config"]["rccl_ranks"] == 0 and d["config"]["dist_backend"] == "gloo"   # (a gloo rehearsal; under RCCL: the world size)
    assert d["config"]["value_contract_span"] <= d["config"]["value_own_span"] * (1 + 1e-5)   # (the barrier only adds to the span)
    assert "scale_prediction" not in d and "configs_compact" not in d                 # (N = 1 only)
    assert d["roofline"]["kernel"].startswith("k_collect2 (20 plies per launch)"), d["roofline"]["kernel"]
    assert len(full["detail"]["kernel_us_per_rank"]) == ranks and all(u > 0 for u in full["detail"]["kernel_us_per_rank"])
    assert d["config"]["kernel_us_max"] >= d["config"]["kernel_us_min"] > 0
    assert d["ms_per_step"] >= d["config"]["ms_per_step_own_span"] * (1 - 1e-5) > 0
    assert all(pl is not None and pl.get("ratio", 0) > 0 for pl in full["detail"]["trajectory_placement_per_rank"])


def test_bench_more_ranks_than_gpus_fails_fast(G):
    """The driver's SCALE run pointed at a box with fewer GPUs than ranks (this one has one): `--gpus 8`, bare and under a
    launcher, ends at once with a message naming the cause -- no rank reaches the rendezvous, none touches the card."""
    import subprocess
    import sys
    import time
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    t0 = time.time()
    bare = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "8", "--steps", "20", "--warmup", "5"],
                          capture_output=True, text=True, timeout=200, env=env, cwd=root)
    assert bare.returncode != 0 and "--gpus 8 but this node has 1 GPU(s)" in bare.stderr and bare.stdout.strip() == ""
    launched = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr",
                               "127.0.0.1", "--master-port", "29519", os.path.join(root, "bench.py"), "--gpus", "8", "--steps", "20",
                               "--warmup", "5"], capture_output=True, text=True, timeout=300, env=env, cwd=root)
    assert launched.returncode != 0 and "8 rank(s) but this node has 1 GPU(s)" in launched.stderr
    assert time.time() - t0 < 240


def test_collect_beyond_4gib(G):
    """A trajectory whose observation array exceeds 2^32 bytes (2^20 boards x 40 plies x 117 B = 4.9 GB): slot
    offsets must be 64-bit and the buffer descriptors of the streaming stores must reach every tile.  The first, a
    middle and the last slot against the ply-by-ply pipeline, whole tensors."""
    n, T, seed = 1 << 20, 40, 13
    a = G.BatchedGobblet(n, DEV, auto_reset=True, seed=seed)
    b = G.BatchedGobblet(n, DEV, auto_reset=True, seed=seed)
    tr = a.collect(T, refresh=False)
    assert tr["observation"].numel() > (1 << 32)
    for t_ in range(T):
        b.rollout(1)
        if t_ in (0, 17, T - 1):
            assert torch.equal(tr["observation"][t_], b.observation) and torch.equal(tr["action_mask"][t_], b.action_mask), t_
            assert torch.equal(tr["actions"][t_], b.actions) and torch.equal(tr["done"][t_], b.done), t_
    assert torch.equal(a.squares, b.squares) and torch.equal(a.to_move, b.to_move)
    del tr


def test_placement_probe_and_spread_buffers(G):
    """gbl_placement_probe writes zeros into exactly the bytes it was given and reports three positive times;
    trajectory_buffers(placement="auto") at a size it probes hands back zero-filled arrays of the usual shapes, records
    what it did, releases what it only held, and the trajectory collected into them is bit-identical to the one
    collected into buffers placed as the allocator pleases."""
    import ctypes as C
    from gobblet_rl_amd import placement
    nat, L = G._native, G._native.lib()
    guard, na, nb = 4096, 96 << 20, 48 << 20
    a = torch.full((na + 2 * guard,), 7, dtype=torch.uint8, device=DEV)
    b = torch.full((nb + 2 * guard,), 9, dtype=torch.uint8, device=DEV)
    both, ua, ub = C.c_float(), C.c_float(), C.c_float()
    nat.check(L.gbl_placement_probe(a.data_ptr() + guard, na, b.data_ptr() + guard, nb, 0, 0, C.byref(both), C.byref(ua), C.byref(ub),
                                    nat.current_stream(torch.device(DEV))))
    torch.cuda.synchronize()
    assert both.value > 0 and ua.value > 0 and ub.value > 0 and both.value < 2 * (ua.value + ub.value)
    assert int(a[:guard].min()) == 7 and int(a[-guard:].min()) == 7 and int(b[:guard].min()) == 9 and int(b[-guard:].min()) == 9
    tiles = min(na // (4 * 7488), nb // (4 * 3456)) & ~1      # what the probe covers: 4 slots of `tiles` tiles
    assert int(a[guard:guard + 4 * tiles * 7488].max()) == 0 and int(b[guard:guard + 4 * tiles * 3456].max()) == 0
    assert int(a[guard + 4 * tiles * 7488:guard + na].min()) == 7 and int(b[guard + 4 * tiles * 3456:guard + nb].min()) == 9
    rc = L.gbl_placement_probe(a.data_ptr() + 64, na, b.data_ptr(), nb, 0, 0, C.byref(both), C.byref(ua), C.byref(ub), None)
    assert rc == nat.ERR_ALIGN
    rc = L.gbl_placement_probe(a.data_ptr(), 1000, b.data_ptr(), nb, 0, 0, C.byref(both), C.byref(ua), C.byref(ub), None)
    assert rc == nat.ERR_ARG
    # the trajectory's own geometry: 3 slots of 8192 boards; a slot count the buffers cannot hold is refused
    a[guard:guard + na] = 7; b[guard:guard + nb] = 9
    nat.check(L.gbl_placement_probe(a.data_ptr() + guard, na, b.data_ptr() + guard, nb, 8192, 3, C.byref(both), C.byref(ua), C.byref(ub), None))
    torch.cuda.synchronize()
    assert int(a[guard:guard + 3 * 8192 * 117].max()) == 0 and int(a[guard + 3 * 8192 * 117:guard + na].min()) == 7
    assert int(b[guard:guard + 3 * 8192 * 54].max()) == 0 and int(b[guard + 3 * 8192 * 54:guard + nb].min()) == 9
    assert L.gbl_placement_probe(a.data_ptr() + guard, na, b.data_ptr() + guard, nb, 8192, 4000, C.byref(both), C.byref(ua), C.byref(ub), None) == nat.ERR_ARG
    assert L.gbl_placement_probe(a.data_ptr() + guard, na, b.data_ptr() + guard, nb, 8200, 3, C.byref(both), C.byref(ua), C.byref(ub), None) == nat.ERR_ARG
    del a, b

    n, T, seed = 65536, 24, 5     # mask trajectory 81 MiB: large enough to be probed
    kw = dict(auto_reset=True, seed=seed)
    e1, e2 = G.BatchedGobblet(n, DEV, **kw), G.BatchedGobblet(n, DEV, **kw)
    free0, _ = torch.cuda.mem_get_info()
    spread = e1.trajectory_buffers(T)  # placement="auto"
    info = spread["_placement"]
    assert set(info) >= {"spread", "ratio", "probes", "block_gib", "held_gib", "cap_gib", "ended", "seconds"} and 1 <= len(info["probes"]) <= placement.MAX_PROBES + 1 + len(placement.FAR_GAPS_BYTES)  # (+ the allocator's own pair, + the far candidates)
    assert 0.5 < info["ratio"] < 1.2 and info["held_gib"] <= info["cap_gib"] <= placement.MAX_HOLD_BYTES / placement.GIB
    assert info["cap_gib"] * placement.GIB <= max(free0 / placement.FREE_FRACTION, 4 * placement.GIB) + (1 << 30)
    assert spread["observation"].shape == (T, n, 3, 3, 13) and spread["action_mask"].shape == (T, n, 54)
    assert int(spread["_full"]["observation"].abs().max()) == 0 and int(spread["_full"]["action_mask"].abs().max()) == 0
    free1, _ = torch.cuda.mem_get_info()
    assert free0 - free1 < 5 * placement.GIB          # only the two blocks of the arrays stay (2 GiB each); the rest went back
    plain = e2.trajectory_buffers(T, placement="any")
    assert plain["_placement"]["spread"] is False
    e1.collect(T, out=spread); e2.collect(T, out=plain)
    for key in ("actions", "winner", "rewards", "done", "to_move", "action_mask", "observation"):
        assert torch.equal(spread[key], plain[key]), key
    assert torch.equal(e1.squares, e2.squares)
    # inside a graph capture nothing may synchronise: buffers made there are not probed (and the capture survives)
    e3 = G.BatchedGobblet(n, DEV, **kw)
    e3.device_ply()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, capture_error_mode="thread_local"):
            tr = e3.collect(T, out=e3.trajectory_buffers(T), refresh=False)
            e3.advance_ply()
        g.replay()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    assert tr["_placement"]["why"] == "inside a graph capture" and torch.equal(tr["actions"], plain["actions"])
    with pytest.raises(ValueError):
        G.BatchedGobblet(64, DEV, auto_reset=True).trajectory_buffers(4, placement="spread")  # too small to probe
    small = G.BatchedGobblet(64, DEV, auto_reset=True).trajectory_buffers(4)
    assert small["_placement"]["spread"] is False and "too small" in small["_placement"]["why"]
    # implicit buffers: the environment's own staging set, made once per shape (placed like any other: here too small to probe)
    small = G.BatchedGobblet(64, DEV, auto_reset=True)
    first = small.collect(4)
    assert first["_placement"]["why"] == "arrays too small to probe" and small.collect(4) is first and small.collect(5) is not first
    small.release_staging()
    assert small.collect(4) is not first
    # ... at most STAGING_SETS of them are kept (least recently used dropped first), and out="fresh" hands out buffers of the
    # caller's own that no later call overwrites
    four = small.collect(4)
    five = small.collect(5)
    assert small.collect(4) is four and small.collect(6) is not None and len(small._staging) == small.STAGING_SETS == 2
    assert small.collect(5) is not five and small.collect(4) is not four
    fresh = small.collect(4, out="fresh")
    kept = fresh["actions"].clone()
    assert small.collect(4) is not fresh and small.collect(4, out="fresh") is not fresh and torch.equal(fresh["actions"], kept)
    with pytest.raises(ValueError):
        small.collect(4, out="new")
    # reset() KEEPS the staging sets (a captured graph may hold their addresses, and a reset-then-collect loop must not
    # search for a placement every episode); a set that is used while a stream is capturing is pinned against the LRU eviction
    again = small.collect(4)
    small.reset()
    assert small._staging and small.collect(4) is again
    pinned_env = G.BatchedGobblet(64, DEV, auto_reset=True)
    pinned_env.device_ply()
    held = pinned_env.collect(3, refresh=False)          # made (and kept) outside the capture ...
    side2 = torch.cuda.Stream()
    side2.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side2):
        g2 = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g2, capture_error_mode="thread_local"):
            assert pinned_env.collect(3, refresh=False) is held   # ... used inside it: the graph replays into these addresses
            pinned_env.advance_ply()
    torch.cuda.current_stream().wait_stream(side2)
    assert held.get("_pinned") is True
    ptr = held["_full"]["observation"].data_ptr()
    for T_other in (4, 5, 6, 7):
        pinned_env.collect(T_other, refresh=False)       # four more shapes: the LRU evicts among the loose sets only
    pinned_env.reset()
    assert pinned_env.collect(3, refresh=False) is held and held["_full"]["observation"].data_ptr() == ptr
    assert len([v for v in pinned_env._staging.values() if not v.get("_pinned")]) <= pinned_env.STAGING_SETS
    g2.replay()
    torch.cuda.synchronize()
    pinned_env.release_staging()
    assert not pinned_env._staging
    # place(): buffers made WITHOUT the search are probed, or re-homed, afterwards; their contents survive either
    e4 = G.BatchedGobblet(n, DEV, **kw)
    mine = e4.trajectory_buffers(T, placement="any")
    e4.collect(T, out=mine)
    rec = e4.place(mine, rehome=False)
    assert rec["ended"] == "probed only" and 0.5 < rec["ratio"] < 1.2 and mine["_placement"] is rec
    for key in ("action_mask", "observation"):
        assert torch.equal(mine[key], plain[key]), key
    rec2 = e4.place(mine)
    assert "probes" in rec2 and rec2["ratio"] <= rec["ratio"] + 0.08 and mine["observation"].shape == plain["observation"].shape
    for key in ("actions", "action_mask", "observation"):
        assert torch.equal(mine[key], plain[key]), key
    e4.collect(T, out=mine); e2.collect(T, out=plain)
    for key in ("actions", "action_mask", "observation"):
        assert torch.equal(mine[key], plain[key]), key


def test_placement_on_a_nearly_full_device(G):
    """A caller that has filled HBM already (a trainer's model and replay buffer): the search is capped by a quarter of
    what is free, never flushes torch's allocator cache, frees what it only held, and when not even the arrays' own
    blocks fit the buffers are plain torch allocations with the reason recorded -- the trajectories are the same."""
    from gobblet_rl_amd import placement
    GIB = placement.GIB
    n, T, seed = 65536, 24, 5
    kw = dict(auto_reset=True, seed=seed)
    ref = G.BatchedGobblet(n, DEV, **kw)
    plain = ref.trajectory_buffers(T, placement="any")
    ref.collect(T, out=plain)
    cached = torch.empty(1 << 30, dtype=torch.uint8, device=DEV)   # a block that sits in torch's cache during the search
    del cached
    reserved0 = torch.cuda.memory_reserved()
    assert reserved0 >= 1 << 30
    free, _ = torch.cuda.mem_get_info()
    hog = torch.empty(free - 30 * GIB, dtype=torch.uint8, device=DEV)   # leave ~30 GiB: the cap becomes ~7.5 GiB
    e1 = G.BatchedGobblet(n, DEV, **kw)
    tr = e1.trajectory_buffers(T)
    info = tr["_placement"]
    assert "probes" in info and info["cap_gib"] <= 30 / placement.FREE_FRACTION + 0.5 and info["held_gib"] <= max(info["cap_gib"], 4.0)
    assert torch.cuda.memory_reserved() >= reserved0 + hog.numel()      # torch's cache was not flushed
    free1, _ = torch.cuda.mem_get_info()
    assert free1 >= 30 * GIB - 5 * GIB                                   # the rejected blocks and gaps went back to the driver
    e1.collect(T, out=tr)
    for key in ("actions", "winner", "action_mask", "observation"):
        assert torch.equal(tr[key], plain[key]), key
    del tr, e1
    free2, _ = torch.cuda.mem_get_info()
    hog2 = torch.empty(max(free2 - 6 * GIB, 1), dtype=torch.uint8, device=DEV)   # ~6 GiB left: < 2 + 2 + 4 GiB reserve
    e2 = G.BatchedGobblet(n, DEV, **kw)
    tr = e2.trajectory_buffers(T)   # no room for blocks: the pair as torch places it, probed, the reason recorded
    assert tr["_placement"]["block_gib"] == [0.0, 0.0] and len(tr["_placement"]["probes"]) == 1
    assert tr["_placement"]["ended"].startswith(("no block search", "the allocator's own placement is clean"))
    e2.collect(T, out=tr)
    for key in ("actions", "winner", "action_mask", "observation"):
        assert torch.equal(tr[key], plain[key]), key
    del hog, hog2, tr
    torch.cuda.empty_cache()


@pytest.mark.parametrize("n,illegal,auto_reset", [(4099, "noop", True), (300, "terminate", False)])
def test_step_into_trajectory_slots(G, n, illegal, auto_reset):
    """step_into(): an external policy's ply written straight into slot t of the trajectory buffers == step() followed
    by copies, slot by slot (the policy here: the library's sampler on the previous slot's mask, and a few illegal
    actions thrown in)."""
    T, seed = 9, 3
    kw = dict(auto_reset=auto_reset, seed=seed, illegal_mode=illegal, track_turn=True)
    a, b = G.BatchedGobblet(n, DEV, **kw), G.BatchedGobblet(n, DEV, **kw)
    out = a.trajectory_buffers(T)
    mask = a.action_mask
    for t_ in range(T):
        acts = b.sample_actions().clone()       # from b's current mask (== the mask a's policy sees)
        assert torch.equal(mask, b.action_mask)
        acts[::7] = (acts[::7] + 11) % 60        # some illegal / out-of-range actions
        obs_t, mask = a.step_into(acts, out, t_)
        b.step(acts)
        assert torch.equal(out["action_mask"][t_], b.action_mask) and torch.equal(out["observation"][t_], b.observation), t_
        assert torch.equal(out["winner"][t_], b.winner) and torch.equal(out["rewards"][t_], b.rewards), t_
        assert torch.equal(out["done"][t_], b.done) and torch.equal(out["to_move"][t_], b.to_move), t_
        assert torch.equal(out["actions"][t_], acts) and obs_t.data_ptr() == out["observation"][t_].data_ptr()
    assert torch.equal(a.squares, b.squares) and torch.equal(a.turn, b.turn) and a.ply == b.ply
    pad = out["_full"]["observation"][:, n:]
    assert int(pad.abs().sum()) == 0
    with pytest.raises(IndexError):
        a.step_into(acts, out, T)
    with pytest.raises(ValueError):
        a.step_into(acts, a.trajectory_buffers(2, layout="tile"), 0)
