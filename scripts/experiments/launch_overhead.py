#!/usr/bin/env python3
"""What the driver's 20 timed plies cost beyond their kernel (one gbl_collect launch of 20 plies at 2^20 boards: ~545 us of
kernel time): wall clock from a synchronised start to a synchronised end, for the ways of launching and of waiting.
    python scripts/experiments/launch_overhead.py [BOARDS] [PLIES]"""
import os
import statistics
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402
import gobblet_rl_amd as G  # noqa: E402

if os.environ.get("GOBBLET_HIP_LIB"):  # an experiment's own build of the library (scripts/build_variant.sh)
    G._native.use_library(os.environ["GOBBLET_HIP_LIB"])

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
K = int(sys.argv[2]) if len(sys.argv) > 2 else 20
dev = torch.device("cuda:0")
p = bench.Pipeline(G, torch, n, 0, dev, mode="collect", traj=K)
p.eager(5)
g = p.capture(K)
g.replay()
torch.cuda.synchronize(dev)
s = p.nat.current_stream(dev)


def timed(launch, wait):
    res, kern = [], []
    for _ in range(12):
        torch.cuda.synchronize(dev)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        a.record()
        launch()
        b.record()
        wait(b)
        torch.cuda.synchronize(dev)
        res.append((time.perf_counter() - t0) * 1e6)
        kern.append(a.elapsed_time(b) * 1e3)
        p.advance(K, s)  # (bookkeeping for the next launch: outside the timed region)
    return statistics.median(res), statistics.median(kern)


def eager_no_advance():
    p.enqueue(0, K, s)


def spin(ev):
    while not ev.query():
        pass


for lname, launch in (("hipGraph replay (kernel + counter node)", g.replay), ("eager launch, counter bumped afterwards", eager_no_advance)):
    for wname, wait in (("torch.cuda.synchronize", lambda ev: None), ("spin on event.query, then synchronize", spin)):
        wall, kern = timed(launch, wait)
        print(f"{n} boards x {K} plies, {lname}; {wname}: wall {wall:7.1f} us, events {kern:7.1f} us, beyond the events {wall - kern:5.1f} us", flush=True)
