#!/usr/bin/env python3
"""Does the per-ply time of a hipGraph of gbl_collect (or gbl_rollout) launches change with how long the GPU has been
busy?  Replays the same graph back to back and prints every replay's time: a ramp means the chip's clocks (fabric /
memory DPM) need sustained load to come up, i.e. short measurements understate the steady rate."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gobblet_rl_amd as G  # noqa: E402

if os.environ.get("GOBBLET_HIP_LIB"):  # an experiment's own build of the library (scripts/build_variant.sh)
    G._native.use_library(os.environ["GOBBLET_HIP_LIB"])

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
T = int(sys.argv[2]) if len(sys.argv) > 2 else 8
launches = int(sys.argv[3]) if len(sys.argv) > 3 else 3
env = G.BatchedGobblet(n, "cuda:0", auto_reset=True, seed=0)
env.rollout(5)
env.device_ply()
buf = env.trajectory_buffers(T) if T else None
g = torch.cuda.CUDAGraph()
side = torch.cuda.Stream()
with torch.cuda.stream(side):
    if T:
        env.collect(T, out=buf, refresh=False)
    else:
        env.rollout(1)
    env.advance_ply()
    torch.cuda.synchronize()
    with torch.cuda.graph(g, capture_error_mode="thread_local"):
        for _ in range(launches):
            if T:
                env.collect(T, out=buf, refresh=False)
            else:
                env.rollout(1)
        env.advance_ply()
torch.cuda.synchronize()
time.sleep(0.5)  # let the chip go idle
plies = launches * max(T, 1)
out = []
t_start = time.perf_counter()
for i in range(400):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); g.replay(); b.record()
    b.synchronize()
    out.append((time.perf_counter() - t_start, a.elapsed_time(b) * 1e3 / plies))
for i in (0, 1, 2, 3, 5, 8, 12, 20, 30, 50, 80, 120, 200, 300, 399):
    print(f"replay {i:3d} at {out[i][0] * 1e3:8.2f} ms: {out[i][1]:7.2f} us/ply")
