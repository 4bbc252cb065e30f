#!/usr/bin/env python3
"""Second placement experiment (see placement_map.py): a large arena, which offsets are 'the same region' as which, and
does ONE stream gain when its plies alternate between regions.
usage: placement_map2.py [arena GiB]"""
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import gobblet_rl_amd as G  # noqa: E402

if os.environ.get("GOBBLET_HIP_LIB"):  # an experiment's own build of the library (scripts/build_variant.sh)
    G._native.use_library(os.environ["GOBBLET_HIP_LIB"])

n, T = 1 << 20, 8
gib = int(sys.argv[1]) if len(sys.argv) > 1 else 224
nat, L = G._native, G._native.lib()
dev = torch.device("cuda:0")
env = G.BatchedGobblet(n, "cuda:0", auto_reset=True, seed=0)
env.rollout(64)
slot = n
arena = torch.empty(gib << 30, dtype=torch.uint8, device=dev)
base = arena.data_ptr()
ctr = torch.zeros(1, dtype=torch.int32, device=dev)
print(f"arena {gib} GiB @ {base:#x}", flush=True)
GiB = 1 << 30


def measure(obs_off, mask_off, T=T, ply_stride=slot, reps=3, small=None):
    launches = max(2, 64 // T)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        s = nat.current_stream(dev)
        for i in range(launches):
            nat.check(L.gbl_collect(env.squares.data_ptr(), env.to_move.data_ptr(), env.done.data_ptr(),
                                    None, None, None, None, None,
                                    None if mask_off is None else base + mask_off, None if obs_off is None else base + obs_off,
                                    n, ply_stride, 64, 0, 0, i * T, ctr.data_ptr(), T, 0, None, None, s))
        nat.check(L.gbl_counter_add(ctr.data_ptr(), launches * T, s))
    g.replay()
    torch.cuda.synchronize()
    us = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); g.replay(); b.record()
        torch.cuda.synchronize()
        us.append(a.elapsed_time(b) * 1e3 / (launches * T))
    return statistics.median(us)


obs_bytes, mask_bytes = T * slot * 117, T * slot * 54
print("# obs fixed, mask swept in 4 GiB steps: us per ply (both streams, no small arrays)")
for obs_off in (0, 36 * GiB, 100 * GiB, 180 * GiB):
    if obs_off + obs_bytes > (gib << 30):
        continue
    row = []
    for off in range(0, (gib << 30) - mask_bytes, 4 * GiB):
        if off < obs_off + obs_bytes and obs_off < off + mask_bytes:
            off += obs_bytes  # just behind the observation trajectory
        row.append(f"{measure(obs_off, off):5.1f}")
    print(f"obs @ {obs_off >> 30:3d} GiB: " + " ".join(row), flush=True)

print("\n# one stream, T = 2: ply 1 lies `d` GiB behind ply 0 (us per ply)")
for what in ("obs", "mask", "both"):
    rowb = 117 if what != "mask" else 54
    for d in (1, 8, 24, 36, 48, 64, 96):
        stride = (d * GiB // rowb) // 16 * 16  # boards
        far = stride * 117 + slot * 117
        if what == "obs":
            us = measure(0, None, T=2, ply_stride=stride)
        elif what == "mask":
            us = measure(None, 0, T=2, ply_stride=stride)
        else:  # both: observation plies at 0 and d GiB, mask plies from 110 GiB on
            if 110 * GiB + stride * 54 + slot * 54 > (gib << 30) or far > 110 * GiB:
                continue
            us = measure(0, 110 * GiB, T=2, ply_stride=stride)
        print(f"{what:5s} ply distance {d:3d} GiB: {us:6.2f}", flush=True)

print("\n# region boundaries: obs at 0, mask swept in 1 GiB steps over 24..72 GiB")
row = []
for off in range(24 * GiB, 72 * GiB, GiB):
    row.append(f"{measure(0, off):5.1f}")
print(" ".join(row), flush=True)
