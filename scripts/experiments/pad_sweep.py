#!/usr/bin/env python3
"""gbl_collect's time per ply against the padding of a trajectory slot (BatchedGobblet.trajectory_buffers(pad_boards=...)): do the
ply slots of a large batch alias in HBM (their starts are 117 x 2^20 / 2^22 bytes apart: multiples of a large power of two)?
    python scripts/experiments/pad_sweep.py BOARDS T"""
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import gobblet_rl_amd as G  # noqa: E402

if os.environ.get("GOBBLET_HIP_LIB"):  # an experiment's own build of the library (scripts/build_variant.sh)
    G._native.use_library(os.environ["GOBBLET_HIP_LIB"])

n, T = int(sys.argv[1]), int(sys.argv[2])
env = G.BatchedGobblet(n, "cuda:0", auto_reset=True, seed=0)
env.rollout(64)
env.device_ply()
for pad in (0, 128, 384, 1152, 4224, 0, 128):
    buf = env.trajectory_buffers(T, pad_boards=pad)
    g = torch.cuda.CUDAGraph()
    env.collect(T, out=buf, refresh=False); env.advance_ply()
    with torch.cuda.graph(g):
        for _ in range(6):
            env.collect(T, out=buf, refresh=False)
            env.advance_ply()
    g.replay(); torch.cuda.synchronize()
    us = []
    for _ in range(5):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); g.replay(); b.record(); torch.cuda.synchronize()
        us.append(a.elapsed_time(b) * 1e3 / (6 * T))
    print(f"boards {n} T {T} pad {pad:5d}: {statistics.median(us):8.2f} us per ply   placement ratio {buf['_placement'].get('ratio')}", flush=True)
    del g, buf
