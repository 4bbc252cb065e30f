#!/usr/bin/env python3
"""How stable is the per-ply time of gbl_collect across re-allocations of the trajectory buffers within one process,
and how does it depend on the padding between trajectory slots (power-of-two slot strides alias in the memory
channels)?    python scripts/collect_variance.py [BOARDS] [PADS comma separated] [T list]"""
import os
import sys
import statistics

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gobblet_rl_amd as G  # noqa: E402

if os.environ.get("GOBBLET_HIP_LIB"):  # an experiment's own build of the library (scripts/build_variant.sh)
    G._native.use_library(os.environ["GOBBLET_HIP_LIB"])

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
pads = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "0,16,208,592,4112").split(",")]
Ts = [int(x) for x in (sys.argv[3] if len(sys.argv) > 3 else "8,32").split(",")]


def measure(T, launches, pad, keep):
    env = G.BatchedGobblet(n, "cuda:0", auto_reset=True, seed=0)
    env.rollout(64)
    env.device_ply()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    buf = env.trajectory_buffers(T, pad_boards=pad)
    env.collect(T, out=buf, refresh=False)
    torch.cuda.synchronize()
    with torch.cuda.graph(g):
        for _ in range(launches):
            env.collect(T, out=buf, refresh=False)
        env.advance_ply()
    keep.append(buf)
    g.replay()
    torch.cuda.synchronize()
    us = []
    for _ in range(3):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); g.replay(); b.record()
        torch.cuda.synchronize()
        us.append(a.elapsed_time(b) * 1e3 / (T * launches))
    return statistics.median(us)


for T in Ts:
    for pad in pads:
        keep, res = [], []
        for rnd in range(4):
            res.append(measure(T, max(2, 256 // T), pad, keep))
            if rnd % 2:
                keep.clear()
        print(f"boards {n} T={T:2d} pad {pad:6d} boards: " + " ".join(f"{x:7.2f}" for x in res)
              + f"   min {min(res):.2f} max {max(res):.2f} us/ply", flush=True)
