#!/usr/bin/env python3
"""How to reach memory of another class from a fresh process: separate 8 GiB spacers (untouched / zeroed), or one big
arena probed inside.  usage: placement_walk.py empty|zeros|arena [GiB]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gobblet_rl_amd as G  # noqa: E402

if os.environ.get("GOBBLET_HIP_LIB"):  # an experiment's own build of the library (scripts/build_variant.sh)
    G._native.use_library(os.environ["GOBBLET_HIP_LIB"])
from gobblet_rl_amd import placement  # noqa: E402

how = sys.argv[1]
total = int(sys.argv[2]) if len(sys.argv) > 2 else 160
dev = torch.device("cuda:0")
GiB = 1 << 30
n, T = 1 << 20, 8
env = G.BatchedGobblet(n, "cuda:0", auto_reset=True, seed=0)
obs = torch.zeros(T * n * 117, dtype=torch.uint8, device=dev)
print(f"{how}: obs @ {obs.data_ptr():#x}", flush=True)
held = []
t0 = time.perf_counter()
if how == "arena":
    arena = torch.empty(total * GiB, dtype=torch.uint8, device=dev)
    print(f"arena @ {arena.data_ptr():#x} after {time.perf_counter() - t0:.3f} s")
    for off in range(0, total * GiB, 8 * GiB):
        m = arena[off:off + T * n * 54]
        b, ua, ub = placement.probe(obs, m)
        print(f"  +{off // GiB:3d} GiB  ratio {b / (ua + ub):.3f}  (both {b:.0f} a {ua:.0f} b {ub:.0f} us)", flush=True)
else:
    for i in range(total // 8):
        m = torch.zeros(T * n * 54, dtype=torch.uint8, device=dev)
        b, ua, ub = placement.probe(obs, m)
        print(f"  after {8 * i:3d} GiB of spacers: mask @ {m.data_ptr():#x} ratio {b / (ua + ub):.3f}  t = {time.perf_counter() - t0:.3f} s", flush=True)
        held.append(m)
        held.append((torch.zeros if how == "zeros" else torch.empty)(8 * GiB, dtype=torch.uint8, device=dev))
free, tot = torch.cuda.mem_get_info(dev)
print(f"free {free / GiB:.1f} of {tot / GiB:.1f} GiB")
