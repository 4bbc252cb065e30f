"""GPU parity tests (-m gpu) of gbl_collect_policy (k_collect_policy<W>): trajectories collected with a device-side policy
per side -- masked-random or GreedyGobbletPolicy.compute_action at depth 1 / 2 (greedy_policy.py:38-221) -- against the
oracle's restatement of the reference's own game loops (tutorials/GreedyAgent/tutorial_greedy.py:16-54: one greedy policy
object for both agents, random opening plies; examples/example_basic.py:50-67: masked-random), ply by ply: the action, how
it was arrived at (greedy choice / fallback draw / random), the chosen-before-fallback action, the candidate set, the
histories, winner, rewards, done, next mover, mask, observation, the state and the turn counters.  Bit-exact."""
import numpy as np
import pytest
import torch

import oracle

pytestmark = pytest.mark.gpu

DEV = "cuda:0"
THREADS = 16
POL = {"random": 0, "greedy1": 1, "greedy": 2, "greedy2": 2, "greedy3": 3}


@pytest.fixture(scope="module")
def G():
    import gobblet_rl_amd as g
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    g._native.lib()
    return g


def npy(x):
    return x.cpu().numpy()


def run_and_check(G, n, T, policies, opening, seed=5, base=0, warm=6, with_obs=True, illegal="noop", layout="time",
                  device_ply=False, launches=1, candidates=True):
    env = G.BatchedGobblet(n, DEV, auto_reset=True, seed=seed, env_base=base, with_observation=with_obs,
                           illegal_mode=illegal, track_turn=True)
    s, tm, dn = oracle.batch_reset(n)
    turn = np.zeros(n, np.int32)
    if warm:  # staggered game phases, masked-random
        env.rollout(warm)
        oracle.batch_rollout(s, tm, dn, seed, base, 0, warm, threads=THREADS, want_obs=False, want_mask=False, turn=turn)
        assert np.array_equal(npy(env.turn), turn)
    rng = np.random.default_rng(seed)
    hist = rng.integers(-1, 54, (n, 2, 3)).astype(np.int8)  # histories that matter from the first ply on
    env.reset_policy_history()
    env.policy_hist.copy_(torch.from_numpy(hist).to(DEV))
    if device_ply:
        env.device_ply()
    pol = tuple(POL[p] for p in policies)
    im = 0 if illegal == "noop" else 1
    buf = env.trajectory_buffers(T, layout=layout, placement="any", policy_outputs=True, candidates=candidates)
    ply = warm
    seen = {0: 0, 1: 0, 2: 0}
    for _ in range(launches):
        env.collect(T, out=buf, policies=policies, opening_plies=opening, count=True, refresh=False)
        env.advance_ply()
        torch.cuda.synchronize()

        def slot(key, t_):
            if layout == "time":
                return npy(buf[key][t_])
            v = buf[key][:, t_]
            return npy(v.reshape((v.shape[0] * 64,) + tuple(v.shape[2:]))[:n])
        for t_ in range(T):
            o = oracle.batch_policy_ply(s, tm, dn, hist, turn, seed, base, ply, pol, opening, illegal_mode=im,
                                        threads=THREADS, want_obs=with_obs)
            assert np.array_equal(slot("how", t_), o["how"]), ("how", t_)
            assert np.array_equal(slot("chosen", t_), o["chosen"]), ("chosen", t_)
            assert np.array_equal(slot("actions", t_), o["actions"]), ("actions", t_)
            if candidates:
                assert np.array_equal(slot("candidates", t_), o["cands"]), ("candidates", t_)
            assert np.array_equal(slot("winner", t_), o["winner"]) and np.array_equal(slot("rewards", t_), o["reward"]), t_
            assert np.array_equal(slot("done", t_), dn) and np.array_equal(slot("to_move", t_), tm), t_
            assert np.array_equal(slot("action_mask", t_), o["mask"]), ("mask", t_)
            if with_obs:
                assert np.array_equal(slot("observation", t_), o["obs"]), ("obs", t_)
            for k in seen:
                seen[k] += int((o["how"] == k).sum())
            ply += 1
        assert np.array_equal(npy(env.squares), s) and np.array_equal(npy(env.to_move), tm) and np.array_equal(npy(env.done), dn)
        assert np.array_equal(npy(env.turn), turn) and np.array_equal(npy(env.policy_hist), hist)
    assert env.ply == ply
    return seen, env


@pytest.mark.parametrize("n,T,policies,opening,kw", [
    (1, 9, ("greedy", "greedy"), 0, {}),
    (65, 14, ("greedy", "random"), 0, {"layout": "tile"}),
    (4099, 13, ("random", "greedy"), 0, {"illegal": "terminate", "with_obs": False}),
    (3000, 16, ("greedy", "greedy"), 2, {"launches": 2, "device_ply": True}),          # tutorial_greedy.py: 2 random opening plies
    (2000, 12, ("greedy1", "greedy"), 0, {}),                                           # RLlib adapter's depth 1 vs depth 2
    (2000, 12, ("greedy1", "greedy1"), 1, {"warm": 0}),                                 # no pooled rounds: the one-wavefront kernel
    (400, 8, ("greedy3", "random"), 0, {"candidates": False}),
    (1000, 8, ("random", "random"), 0, {}),
    (40000, 6, ("greedy", "greedy"), 0, {"base": 7_000_000_000}),                       # <4,8> blocks (one generation)
    (20037, 5, ("greedy", "greedy"), 0, {"warm": 6}),                                   # <2,8> blocks
    (98341, 3, ("greedy", "greedy"), 0, {"warm": 6, "candidates": False}),              # <1,4>: between two generations
    # ragged last tiles (n % 64 != 0), greedy vs greedy, TERMINATE mode: no valid board needs a masked-random pick, so the lanes
    # past the end take the kernel's cheap path (they must still play a legal move and stay out of every tally)
    (4099 + 37, 9, ("greedy", "greedy"), 0, {"illegal": "terminate", "launches": 2}),
    (70001, 4, ("greedy", "greedy"), 0, {"illegal": "terminate", "candidates": False, "layout": "tile"}),
])
def test_policy_collect_vs_oracle(G, n, T, policies, opening, kw):
    seen, _ = run_and_check(G, n, T, policies, opening, **kw)
    if "greedy" in policies or "greedy3" in policies:
        assert seen[1] > 0 and (n < 100 or seen[2] > 0)       # greedy choices and fallback draws both occurred
    if "random" in policies or opening:
        assert seen[0] > 0 or n == 1


def test_policy_collect_config5_size_selfplay(G):
    """The benchmark's greedy record at its size: 65 536 boards, greedy (depth 2) against greedy, 12 plies in one launch,
    every ply against the oracle (batch_greedy_act + batch_step: decisions, fallback draws and histories included)."""
    seen, env = run_and_check(G, 65536, 12, ("greedy", "greedy"), 0, warm=8, candidates=False)
    assert seen[0] == 0 and seen[1] > 0 and seen[2] > 0
    c = npy(env.counters)
    assert c[0] == 65536 * 12 and c[1] == c[2] + c[3] and c[1] > 0


def test_policy_collect_equals_random_collect_and_stepwise_policy(G):
    """Two cross-checks inside the library: (random, random) leaves exactly gbl_collect's trajectory, and a greedy side
    leaves what GreedyGobbletPolicy (gbl_greedy_act, one launch per ply, the call index set to the ply) + step() leave."""
    n, T, seed = 5000, 10, 8
    kw = dict(auto_reset=True, seed=seed, track_turn=True)
    a, b = G.BatchedGobblet(n, DEV, **kw), G.BatchedGobblet(n, DEV, **kw)
    ta = a.collect(T, policies=("random", "random"))
    tb = b.collect(T)
    for key in ("actions", "winner", "rewards", "done", "to_move", "action_mask", "observation"):
        assert torch.equal(ta[key], tb[key]), key
    assert torch.equal(a.squares, b.squares) and torch.equal(a.turn, b.turn)
    assert int(ta["how"].abs().sum()) == 0 and int((ta["chosen"] != -1).sum()) == 0
    # greedy for player_1, random for player_2, ply by ply with the host-side policy class
    c, d = G.BatchedGobblet(n, DEV, **kw), G.BatchedGobblet(n, DEV, **kw)
    tc = c.collect(T, policies=("greedy", "random"))
    pol = G.GreedyGobbletPolicy(depth=2, seed=seed, device=DEV)
    for t_ in range(T):
        pol._calls = d.ply                                     # (the fallback draw is keyed by the ply index)
        g_act = pol.compute_actions_from_state(d.squares, d.to_move)
        pol.prev_actions[:, 1] = -1                            # (the class also acted for player_2's boards: not part of this game)
        r_act = d.sample_actions().clone()
        act = torch.where(d.to_move == 0, g_act, r_act)
        d.step(act)
        assert torch.equal(tc["actions"][t_], act), t_
        assert torch.equal(tc["action_mask"][t_], d.action_mask) and torch.equal(tc["observation"][t_], d.observation), t_
    assert torch.equal(c.squares, d.squares)
    assert torch.equal(c.policy_hist[:, 0], pol.prev_actions[:, 0]) and int((c.policy_hist[:, 1] != -1).sum()) == 0


def test_policy_collect_argument_errors(G):
    nat, L = G._native, G._native.lib()
    env = G.BatchedGobblet(256, DEV, auto_reset=True)
    with pytest.raises(ValueError):
        env.collect(4, policies=("greedy", "minimax"))
    with pytest.raises(ValueError):
        env.collect(4, policies=("greedy", "greedy"), opening_plies=2)        # no turn counter
    z = torch.zeros(4 * 256, dtype=torch.int32, device=DEV)
    args = lambda p0, p1, opening, turn: (env.squares.data_ptr(), env.to_move.data_ptr(), env.done.data_ptr(), None,  # noqa: E731
                                          z.data_ptr(), None, None, None, None, None, None, None, None, None, 256, 256, 64, 0, 0, 0,
                                          None, 4, p0, p1, opening, 0, None, turn, None)
    assert L.gbl_collect_policy(*args(4, 0, 0, None)) == nat.ERR_ARG
    assert L.gbl_collect_policy(*args(0, -1, 0, None)) == nat.ERR_ARG
    assert L.gbl_collect_policy(*args(2, 2, 2, None)) == nat.ERR_ARG and b"turn" in L.gbl_last_error()
    assert L.gbl_collect_policy(*args(2, 2, 0, None)) == 0                    # hist = NULL: empty histories, nothing written back
    torch.cuda.synchronize()
