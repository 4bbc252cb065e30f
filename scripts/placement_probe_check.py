#!/usr/bin/env python3
"""Does gbl_placement_probe see what gbl_collect feels?  One arena; the observation trajectory at a fixed offset, the
mask trajectory swept; per placement the probe's three times and its ratio both / (a + b), and gbl_collect's time per ply.
usage: placement_probe_check.py [arena GiB] [boards] [T]"""
import ctypes as C
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gobblet_rl_amd as G  # noqa: E402

if os.environ.get("GOBBLET_HIP_LIB"):  # an experiment's own build of the library (scripts/build_variant.sh)
    G._native.use_library(os.environ["GOBBLET_HIP_LIB"])

gib = int(sys.argv[1]) if len(sys.argv) > 1 else 224
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 20
T = int(sys.argv[3]) if len(sys.argv) > 3 else 8
nat, L = G._native, G._native.lib()
dev = torch.device("cuda:0")
env = G.BatchedGobblet(n, "cuda:0", auto_reset=True, seed=0)
env.rollout(64)
slot = -(-n // 128) * 128
arena = torch.empty(gib << 30, dtype=torch.uint8, device=dev)
base = arena.data_ptr()
ctr = torch.zeros(1, dtype=torch.int32, device=dev)
GiB = 1 << 30
obs_bytes, mask_bytes = T * slot * 117, T * slot * 54
print(f"arena {gib} GiB @ {base:#x}; boards {n}, T {T}: obs {obs_bytes >> 20} MiB, mask {mask_bytes >> 20} MiB", flush=True)


def collect_us(obs_off, mask_off, reps=3):
    launches = max(2, 64 // T)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        s = nat.current_stream(dev)
        for i in range(launches):
            nat.check(L.gbl_collect(env.squares.data_ptr(), env.to_move.data_ptr(), env.done.data_ptr(), None, None, None, None, None,
                                    base + mask_off, base + obs_off, n, slot, 64, 0, 0, i * T, ctr.data_ptr(), T, 0, None, None, s))
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


def probe(obs_off, mask_off):
    both, ua, ub = C.c_float(), C.c_float(), C.c_float()
    nat.check(L.gbl_placement_probe(base + obs_off, obs_bytes, base + mask_off, mask_bytes, slot, T, C.byref(both), C.byref(ua), C.byref(ub),
                                    nat.current_stream(dev)))
    return both.value, ua.value, ub.value


for obs_off in (0, 100 * GiB):
    if obs_off + obs_bytes > gib * GiB:
        continue
    print(f"# obs @ {obs_off >> 30} GiB")
    for off in range(0, gib * GiB - mask_bytes, 8 * GiB):
        if off < obs_off + obs_bytes and obs_off < off + mask_bytes:
            off = obs_off + obs_bytes
        b, ua, ub = probe(obs_off, off)
        print(f"mask @ {off / GiB:6.1f} GiB  probe both {b:7.1f} a {ua:7.1f} b {ub:7.1f} us  ratio {b / (ua + ub):.3f}   "
              f"gbl_collect {collect_us(obs_off, off):6.2f} us/ply", flush=True)
