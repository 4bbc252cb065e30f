#!/usr/bin/env python3
"""Where in device memory do the trajectory arrays of gbl_collect want to lie?  One arena of G GiB (one hipMalloc), the
observation and the mask trajectory placed at chosen offsets inside it, the same launch timed at every placement:
   obs only / mask only over the arena in 1 GiB steps, a plain fill_ of the same bytes beside it,
   both streams with one fixed and the other swept.
usage: placement_map.py [boards] [T] [arena GiB]"""
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import gobblet_rl_amd as G  # noqa: E402

if os.environ.get("GOBBLET_HIP_LIB"):  # an experiment's own build of the library (scripts/build_variant.sh)
    G._native.use_library(os.environ["GOBBLET_HIP_LIB"])

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
T = int(sys.argv[2]) if len(sys.argv) > 2 else 8
gib = int(sys.argv[3]) if len(sys.argv) > 3 else 48
nat, L = G._native, G._native.lib()
dev = torch.device("cuda:0")
env = G.BatchedGobblet(n, "cuda:0", auto_reset=True, seed=0)
env.rollout(64)
slot = -(-n // 128) * 128
small = {k: torch.zeros(T * slot * b, dtype=torch.uint8, device=dev)
         for k, b in {"actions": 4, "winner": 1, "rewards": 2, "done": 1, "to_move": 1}.items()}
arena = torch.empty(gib << 30, dtype=torch.uint8, device=dev)
base = arena.data_ptr()
assert base % (2 << 20) == 0, hex(base)
obs_bytes, mask_bytes = T * slot * 117, T * slot * 54
ctr = torch.zeros(1, dtype=torch.int32, device=dev)
launches = max(2, 128 // T)
print(f"arena {gib} GiB @ {base:#x}; boards {n} T {T}; obs {obs_bytes / 2**20:.0f} MiB mask {mask_bytes / 2**20:.0f} MiB", flush=True)


def measure(obs_off, mask_off, reps=5):
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        s = nat.current_stream(dev)
        for i in range(launches):
            nat.check(L.gbl_collect(env.squares.data_ptr(), env.to_move.data_ptr(), env.done.data_ptr(),
                                    small["actions"].data_ptr(), small["winner"].data_ptr(), small["rewards"].data_ptr(),
                                    small["done"].data_ptr(), small["to_move"].data_ptr(),
                                    None if mask_off is None else base + mask_off, None if obs_off is None else base + obs_off,
                                    n, slot, 64, 0, 0, i * T, ctr.data_ptr(), T, 0, None, None, s))
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


def fill_us(off, nbytes, reps=5):
    v = arena[off:off + nbytes]
    v.fill_(1)
    torch.cuda.synchronize()
    us = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); v.fill_(1); b.record()
        torch.cuda.synchronize()
        us.append(a.elapsed_time(b) * 1e3)
    return statistics.median(us)


step = 1 << 30
offs = list(range(0, (gib << 30) - obs_bytes - mask_bytes, step))
print("\n# one stream at a time: us per ply (and TB/s), fill_ of the same bytes (TB/s)")
single = {}
for off in offs:
    o = measure(off, None)
    m = measure(None, off)
    f = fill_us(off, obs_bytes)
    single[off] = (o, m)
    print(f"off {off >> 30:3d} GiB  obs {o:6.2f} us ({slot * 117 / o / 1e6:5.2f} TB/s)   mask {m:6.2f} us ({slot * 54 / m / 1e6:5.2f} TB/s)"
          f"   fill_ {obs_bytes / f / 1e6:5.2f} TB/s", flush=True)

print("\n# both streams: obs at a fixed offset, mask swept (us per ply)")
for obs_off in (offs[0], offs[len(offs) // 2]):
    row = []
    for off in offs:
        if abs(off - obs_off) < obs_bytes + mask_bytes and not (off >= obs_off + obs_bytes or off + mask_bytes <= obs_off):
            row.append("   -- ")
            continue
        row.append(f"{measure(obs_off, off, 3):6.2f}")
    print(f"obs @ {obs_off >> 30:3d} GiB: " + " ".join(row), flush=True)
print("\n# both streams back to back (mask directly behind obs), pair swept")
row = []
for off in offs:
    row.append(f"{measure(off, off + obs_bytes, 3):6.2f}")
print(" ".join(row), flush=True)
print("\n# sub-GiB: pair back to back, start swept in 64 MiB steps over the first 2 GiB")
row = []
for off in range(0, 2 << 30, 64 << 20):
    row.append(f"{measure(off, off + obs_bytes, 3):6.2f}")
print(" ".join(row), flush=True)
