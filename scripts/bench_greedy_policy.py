#!/usr/bin/env python3
"""Whole policy steps of the host GreedyGobbletPolicy class (decision + fallback draw + history append) and
greedy-vs-greedy self-play plies (policy step + env step) per second.

    python scripts/bench_greedy_policy.py [boards] [depth]
"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gobblet_rl_amd as G  # noqa: E402

if os.environ.get("GOBBLET_HIP_LIB"):  # an experiment's own build of the library (scripts/build_variant.sh)
    G._native.use_library(os.environ["GOBBLET_HIP_LIB"])

n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
depth = int(sys.argv[2]) if len(sys.argv) > 2 else 2
env = G.BatchedGobblet(n, "cuda:0", auto_reset=True, seed=0)
for _ in range(64):
    env.rollout(1)
pol = G.GreedyGobbletPolicy(depth=depth, device="cuda:0")
for _ in range(3):
    pol.compute_actions_from_state(env.squares, env.to_move)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
iters = 20
e0.record()
for _ in range(iters):
    pol.compute_actions_from_state(env.squares, env.to_move)
e1.record(); torch.cuda.synchronize()
ms_policy = e0.elapsed_time(e1) / iters
e0.record()
for _ in range(iters):
    env.step(pol.compute_actions_from_state(env.squares, env.to_move))
e1.record(); torch.cuda.synchronize()
ms_ply = e0.elapsed_time(e1) / iters
print(json.dumps({"boards": n, "depth": depth, "ms_per_policy_step": ms_policy, "policy_steps_per_s": n / ms_policy * 1e3,
                  "ms_per_selfplay_ply": ms_ply, "selfplay_plies_per_s": n / ms_ply * 1e3}))
