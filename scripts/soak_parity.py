#!/usr/bin/env python3
"""Randomised parity soak on a GPU: random batch sizes / modes / seeds, every kernel against the CPU
oracle, bit for bit, until the time budget is used.  Prints one summary line per round and a total."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gobblet_rl_amd as G  # noqa: E402

if os.environ.get("GOBBLET_HIP_LIB"):  # an experiment's own build of the library (scripts/build_variant.sh)
    G._native.use_library(os.environ["GOBBLET_HIP_LIB"])
import oracle  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
master = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
DEV = "cuda:0"
nat = G._native
t_end = time.time() + budget
rounds = boards_checked = 0


def t(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


while time.time() < t_end:
    n = int(master.choice([1, 7, 63, 64, 65, 1000, 4097, 30011, 131072 + 5, int(master.integers(1, 200000))]))
    seed = int(master.integers(0, 2 ** 62))
    base = int(master.integers(0, 2 ** 40))
    illegal = str(master.choice(["noop", "terminate"]))
    auto = bool(master.integers(0, 2))
    with_obs = bool(master.integers(0, 4) > 0)
    env = G.BatchedGobblet(n, DEV, illegal_mode=illegal, auto_reset=True, with_observation=with_obs, seed=seed,
                           env_base=base)
    s, tm, dn = oracle.batch_reset(n)
    k0 = int(master.integers(1, 40))
    env.rollout(k0, count=True)
    o = oracle.batch_rollout(s, tm, dn, seed, base, 0, k0, illegal_mode=nat.ILLEGAL_NOOP if illegal == "noop" else 1,
                             threads=8)
    assert np.array_equal(env.squares.cpu().numpy(), s) and np.array_equal(env.action_mask.cpu().numpy(), o["mask"])
    assert np.array_equal(env.counters.cpu().numpy(), o["counters"])
    # external-action steps (with wild actions), chosen auto_reset mode
    env.auto_reset = auto
    env.done.zero_(); dn[:] = 0
    env.refresh()
    for k in range(3):
        m = oracle.batch_legal_mask(s, tm)
        a = oracle.batch_sample(m, seed, base, 1000 + k)
        wild = master.random(n) < 0.1
        a = np.where(wild | (a < 0), master.integers(-2, 57, n), a).astype(np.int32)
        exp_status = oracle.batch_action_status(s, tm, dn, a, auto_reset=auto)   # (of the position before the step)
        oo = oracle.batch_step(s, tm, dn, a, illegal_mode=0 if illegal == "noop" else 1, auto_reset=auto, threads=8)
        if k == 1:   # gbl_step_ex: the status byte of every action and the next mover's draw from the mask this launch stores
            st8, nxt = torch.full((n,), 9, dtype=torch.int8, device=DEV), torch.full((n,), 9, dtype=torch.int32, device=DEV)
            ply_next = env.ply + 1
            obs, rew, done, win = env.step(t(a), status=st8, next_actions=nxt)
            assert np.array_equal(st8.cpu().numpy(), exp_status)
            assert np.array_equal(nxt.cpu().numpy(), oracle.batch_sample(oo["mask"], seed, base, ply_next))
        else:
            obs, rew, done, win = env.step(t(a))
        assert np.array_equal(env.squares.cpu().numpy(), s) and np.array_equal(done.cpu().numpy(), dn)
        assert np.array_equal(obs["action_mask"].cpu().numpy(), oo["mask"]) and np.array_equal(win.cpu().numpy(), oo["winner"])
        assert np.array_equal(rew.cpu().numpy(), oo["reward"])
        if with_obs:
            assert np.array_equal(obs["observation"].cpu().numpy(), oo["obs"])
    # a collected trajectory (gbl_collect, time- or tile-major) and a one-launch board evaluation (gbl_board_eval)
    # (collect() plays with auto-reset whatever this round drew for the stepping above: boards the frozen-board steps
    #  left finished are reset on both sides first, then the environment collects with auto_reset on)
    if True:
        if not auto:
            fin = dn != 0
            s[fin] = 0; tm[fin] = 0; dn[:] = 0
            env.auto_reset = True
            env.reset_where(t(fin))
        T = int(master.integers(1, 12))
        lay = str(master.choice(["time", "tile"]))
        ply = env.ply
        tr = env.collect(T, layout=lay)
        torch.cuda.synchronize()
        for tt in range(T):
            oo = oracle.batch_rollout(s, tm, dn, seed, base, ply + tt, 1, illegal_mode=0 if illegal == "noop" else 1, threads=8)
            pick = (lambda k: tr[k][tt]) if lay == "time" else (lambda k: tr[k][:, tt].reshape((-1,) + tuple(tr[k].shape[3:]))[:n])
            assert np.array_equal(pick("actions").cpu().numpy(), oo["actions"]) and np.array_equal(pick("action_mask").cpu().numpy(), oo["mask"])
            assert np.array_equal(pick("winner").cpu().numpy(), oo["winner"]) and np.array_equal(pick("to_move").cpu().numpy(), tm)
            if with_obs:
                assert np.array_equal(pick("observation").cpu().numpy(), oo["obs"])
        assert np.array_equal(env.squares.cpu().numpy(), s)
    # a trajectory with device-side policies (gbl_collect_policy): random sides, depths, opening plies, histories
    if n <= 70000:
        pol = tuple(int(x) for x in master.integers(0, 4, 2))
        opening = int(master.choice([0, 0, 1, 2]))
        T = int(master.integers(1, 7))
        penv = G.BatchedGobblet(n, DEV, illegal_mode=illegal, auto_reset=True, with_observation=with_obs, seed=seed, env_base=base,
                                track_turn=True)
        penv.board.squares = t(s); penv.to_move.copy_(t(tm)); penv.done.zero_()
        turn = master.integers(0, 4, n).astype(np.int32)
        penv.turn.copy_(t(turn)); penv.refresh()
        hist = master.integers(-1, 54, (n, 2, 3)).astype(np.int8)
        penv.reset_policy_history(); penv.policy_hist.copy_(t(hist))
        penv.ply = 5000
        ps, ptm, pdn = s.copy(), tm.copy(), np.zeros(n, np.int8)
        lay = str(master.choice(["time", "tile"]))
        buf = penv.trajectory_buffers(T, layout=lay, placement="any", policy_outputs=True, candidates=True)
        penv.collect(T, out=buf, policies=pol, opening_plies=opening, refresh=False)
        torch.cuda.synchronize()
        for tt in range(T):
            oo = oracle.batch_policy_ply(ps, ptm, pdn, hist, turn, seed, base, 5000 + tt, pol, opening,
                                         illegal_mode=0 if illegal == "noop" else 1, threads=8, want_obs=with_obs)
            pick = (lambda k: buf[k][tt]) if lay == "time" else (lambda k: buf[k][:, tt].reshape((-1,) + tuple(buf[k].shape[3:]))[:n])
            for key, exp in (("actions", oo["actions"]), ("how", oo["how"]), ("chosen", oo["chosen"]), ("candidates", oo["cands"]),
                             ("winner", oo["winner"]), ("done", pdn), ("to_move", ptm), ("action_mask", oo["mask"])):
                assert np.array_equal(pick(key).cpu().numpy(), exp), (key, tt, pol, opening)
            if with_obs:
                assert np.array_equal(pick("observation").cpu().numpy(), oo["obs"])
        assert np.array_equal(penv.squares.cpu().numpy(), ps) and np.array_equal(penv.policy_hist.cpu().numpy(), hist)
        assert np.array_equal(penv.turn.cpu().numpy(), turn)
        del penv, buf
    ev = G.BatchedBoard(n, DEV, squares=t(s)).evaluate()
    assert np.array_equal(ev["winner"].cpu().numpy(), oracle.batch_winner(s)) and np.array_equal(ev["flat"].cpu().numpy(), oracle.batch_flatboard(s))
    assert np.array_equal(ev["mask1"].cpu().numpy(), oracle.batch_legal_mask(s, np.ones(n, np.int8)))
    assert np.array_equal(ev["obs0"].cpu().numpy(), oracle.batch_observe(s, np.zeros(n, np.int8), 0))
    # board API + greedy on the current states
    b = G.BatchedBoard(n, DEV, squares=t(s))
    assert np.array_equal(b.check_for_winner().cpu().numpy(), oracle.batch_winner(s))
    assert np.array_equal(b.get_flatboard().cpu().numpy(), oracle.batch_flatboard(s))
    assert np.array_equal(b.check_covered().cpu().numpy(), oracle.batch_covered(s))
    assert (b.validate(raise_on_error=False) == 0).all()
    m = min(n, 20000)
    hist = master.integers(-1, 54, (m, 2, 3)).astype(np.int8)
    for depth in (1, 2):
        act = torch.empty(m, dtype=torch.int32, device=DEV); cm = torch.empty((m, 54), dtype=torch.int8, device=DEV)
        fb = torch.empty(m, dtype=torch.int8, device=DEV)
        st, who, h = t(s[:m]), t(tm[:m]), t(hist)
        nat.check(nat.lib().gbl_greedy(st.data_ptr(), who.data_ptr(), None, h.data_ptr(), depth, act.data_ptr(),
                                       cm.data_ptr(), fb.data_ptr(), m, None))
        torch.cuda.synchronize()
        og = oracle.batch_greedy(s[:m].copy(), tm[:m].copy(), hist=hist, depth=depth)
        assert np.array_equal(act.cpu().numpy(), og[0]) and np.array_equal(cm.cpu().numpy(), og[1])
        assert np.array_equal(fb.cpu().numpy(), og[2])
    # greedy handed a random subset of the legal moves (guards and early breaks of the depth-1/2 loops)
    sub = (oracle.batch_legal_mask(s[:m], tm[:m]) * (master.random((m, 54)) < master.choice([0.1, 0.5, 0.9]))).astype(np.int8)
    sub = np.ascontiguousarray(sub)
    nat.check(nat.lib().gbl_greedy(st.data_ptr(), who.data_ptr(), t(sub).data_ptr(), h.data_ptr(), 2, act.data_ptr(),
                                   cm.data_ptr(), fb.data_ptr(), m, None))
    torch.cuda.synchronize()
    og = oracle.batch_greedy(s[:m].copy(), tm[:m].copy(), mask=sub, hist=hist, depth=2)
    assert np.array_equal(act.cpu().numpy(), og[0]) and np.array_equal(cm.cpu().numpy(), og[1])
    assert np.array_equal(fb.cpu().numpy(), og[2])
    # whole policy steps (decision + fallback draw + history append), three calls so that histories matter
    hd, ho = t(hist), hist.copy()
    fin = torch.empty(m, dtype=torch.int32, device=DEV)
    for call in range(3):
        nat.check(nat.lib().gbl_greedy_act(st.data_ptr(), who.data_ptr(), None, hd.data_ptr(), 2, seed, base, call,
                                           fin.data_ptr(), act.data_ptr(), cm.data_ptr(), fb.data_ptr(), m, None))
        torch.cuda.synchronize()
        og = oracle.batch_greedy_act(s[:m].copy(), tm[:m].copy(), ho, seed, base, call, depth=2)
        assert np.array_equal(fin.cpu().numpy(), og[0]) and np.array_equal(act.cpu().numpy(), og[1])
        assert np.array_equal(cm.cpu().numpy(), og[2]) and np.array_equal(fb.cpu().numpy(), og[3])
        assert np.array_equal(hd.cpu().numpy(), ho)
    rounds += 1
    boards_checked += n
    print(f"round {rounds}: n={n} illegal={illegal} auto_reset={auto} obs={with_obs} plies={k0}+3 OK", flush=True)
print(f"SOAK OK: {rounds} rounds, {boards_checked} boards, every comparison bit-exact")
