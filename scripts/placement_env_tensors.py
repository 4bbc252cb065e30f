#!/usr/bin/env python3
"""Does the one-ply pipeline (outputs overwritten in place every ply) care where env.observation and env.action_mask lie?
usage: placement_env_tensors.py [boards]"""
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gobblet_rl_amd as G  # noqa: E402

if os.environ.get("GOBBLET_HIP_LIB"):  # an experiment's own build of the library (scripts/build_variant.sh)
    G._native.use_library(os.environ["GOBBLET_HIP_LIB"])
from gobblet_rl_amd import placement  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 22
dev = torch.device("cuda:0")


def timed(env, plies=64):
    env.device_ply()
    env.rollout(8)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(plies):
            env.rollout(1)
        env.advance_ply()
    g.replay()
    torch.cuda.synchronize()
    us = []
    for _ in range(5):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); g.replay(); b.record()
        torch.cuda.synchronize()
        us.append(a.elapsed_time(b) * 1e3 / plies)
    return statistics.median(us)


for rnd in range(3):
    env = G.BatchedGobblet(n, dev, auto_reset=True, seed=0)
    both, ua, ub = placement.probe(env.observation.view(-1), env.action_mask.view(-1))
    env.refresh()
    t_any = timed(env)
    a, b, info = placement.spread_pair(n * 117, n * 54, dev)
    obs, mask = a.view(torch.int8).view(n, 3, 3, 13), b.view(torch.int8).view(n, 54)
    env2 = G.BatchedGobblet(n, dev, auto_reset=True, seed=0)
    env2.observation, env2.action_mask = obs, mask
    env2.refresh()
    t_spread = timed(env2)
    print(f"boards {n} round {rnd}: as allocated (probe ratio {both / (ua + ub):.3f}) {t_any:7.2f} us per ply;  spread "
          f"(ratio {info['ratio']}, probes {info['probes']}) {t_spread:7.2f} us per ply", flush=True)
    keep = env  # keep the first environment alive so that the next round allocates elsewhere
