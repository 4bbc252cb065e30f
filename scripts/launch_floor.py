#!/usr/bin/env python3
"""What ONE dependent kernel launch costs inside a replayed hipGraph, whatever the kernel does: a graph of 200 launches of the
library's one-thread counter kernel (gbl_counter_add), of an EMPTY-board gbl_legal_mask on 64 / 4 096 boards, and of the one-ply
pipeline (gbl_rollout, plies = 1) at 4 096 boards -- the floor under any one-ply-per-launch pipeline at small batches."""
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gobblet_rl_amd as G  # noqa: E402

if os.environ.get("GOBBLET_HIP_LIB"):  # an experiment's own build of the library (scripts/build_variant.sh)
    G._native.use_library(os.environ["GOBBLET_HIP_LIB"])

nat, L = G._native, G._native.lib()
dev = torch.device("cuda:0")
ctr = torch.zeros(1, dtype=torch.int32, device=dev)


def timed(name, body, launches=200):
    body()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(launches):
            body()
    g.replay(); torch.cuda.synchronize()
    us = []
    for _ in range(7):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); g.replay(); b.record(); torch.cuda.synchronize()
        us.append(a.elapsed_time(b) * 1e3 / launches)
    print(f"{name:58s}: {statistics.median(us):6.2f} us per launch", flush=True)


timed("gbl_counter_add (one thread)", lambda: nat.check(L.gbl_counter_add(ctr.data_ptr(), 1, nat.current_stream(dev))))
for n in (64, 4096, 131072):
    env = G.BatchedGobblet(n, dev, auto_reset=True, seed=0)
    env.rollout(16)
    timed(f"gbl_legal_mask, {n} boards (read 27 B, write 54 B per board)",
          lambda: nat.check(L.gbl_legal_mask(env.squares.data_ptr(), env.to_move.data_ptr(), env.action_mask.data_ptr(), n, nat.current_stream(dev))))
    env.device_ply()
    timed(f"gbl_rollout(plies = 1), {n} boards (the one-ply pipeline)", lambda: (env.rollout(1), env.advance_ply()), launches=100)
