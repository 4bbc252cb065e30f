#!/usr/bin/env python3
"""Does a ragged last tile (a batch that is not a multiple of 64 boards) cost more than its share?  Times every batched entry
point whose launch can wait for one slow tile at N and N + 63 boards (hipGraph replays, us per launch):
    gbl_greedy depth 2 | gbl_step (FULL) | gbl_rollout(1) (FULL) | gbl_legal_mask | gbl_observe | gbl_collect_policy (16 plies)
  python scripts/experiments/ragged_sweep.py"""
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import gobblet_rl_amd as G  # noqa: E402

dev = torch.device("cuda:0")
nat, L = G._native, G._native.lib()


def timed(fn, reps=20, inner=10):
    g = torch.cuda.CUDAGraph()
    fn()
    torch.cuda.synchronize()
    with torch.cuda.graph(g):
        for _ in range(inner):
            fn()
    g.replay()
    torch.cuda.synchronize()
    out = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); g.replay(); b.record()
        torch.cuda.synchronize()
        out.append(a.elapsed_time(b) * 1e3 / inner)
    return statistics.median(out)


for base in (4096, 16384, 65536, 262144, 1048576):
    row = {}
    for n in (base, base + 63):
        env = G.BatchedGobblet(n, dev, auto_reset=True, seed=0)
        env.rollout(64)
        s = lambda: nat.current_stream(dev)  # noqa: E731
        act = torch.empty(n, dtype=torch.int32, device=dev)
        cm = torch.empty((n, 54), dtype=torch.int8, device=dev)
        fb = torch.empty(n, dtype=torch.int8, device=dev)
        r = {}
        r["greedy2"] = timed(lambda: nat.check(L.gbl_greedy(env.squares.data_ptr(), env.to_move.data_ptr(), None, None, 2, act.data_ptr(),
                                                            cm.data_ptr(), fb.data_ptr(), n, s()), "gbl_greedy"))
        actions = env.sample_actions().clone()
        r["step"] = timed(lambda: nat.check(L.gbl_step(env.squares.data_ptr(), env.to_move.data_ptr(), env.done.data_ptr(), actions.data_ptr(),
                                                       env.winner.data_ptr(), env.rewards.data_ptr(), env.action_mask.data_ptr(),
                                                       env.observation.data_ptr(), None, n, 0, 1, s()), "gbl_step"))
        r["rollout1"] = timed(lambda: env.rollout(1))
        r["mask"] = timed(lambda: nat.check(L.gbl_legal_mask(env.squares.data_ptr(), env.to_move.data_ptr(), env.action_mask.data_ptr(), n, s()), "m"))
        r["observe"] = timed(lambda: nat.check(L.gbl_observe(env.squares.data_ptr(), env.to_move.data_ptr(), -1, env.observation.data_ptr(), n, s()), "o"))
        if n <= 262144 + 63:
            buf = env.trajectory_buffers(16, policy_outputs=True)
            env.device_ply()
            r["policy16"] = timed(lambda: env.collect(16, out=buf, policies=("greedy", "greedy"), refresh=False), reps=8, inner=2) / 16
        row[n] = r
        del env
    for k in row[base]:
        a, b = row[base][k], row[base + 63][k]
        print(f"boards {base:8d} / +63  {k:9s}: {a:9.2f} / {b:9.2f} us   ({(b / a - 1) * 100:+5.1f} %)", flush=True)
