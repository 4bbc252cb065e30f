#!/usr/bin/env python3
"""Diagnostic (-DGBL_STAMPS build): the shader clock the chip holds while gbl_collect runs (cycle stamps of every
wavefront against the chip-wide 100 MHz clock), next to the launch's write rate -- per CU and clock."""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import gobblet_rl_amd as G  # noqa: E402

if os.environ.get("GOBBLET_HIP_LIB"):  # an experiment's own build of the library (scripts/build_variant.sh)
    G._native.use_library(os.environ["GOBBLET_HIP_LIB"])

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
T = int(sys.argv[2]) if len(sys.argv) > 2 else 8
L = G._native.lib()
L.gbl_debug_stamps.argtypes = [C.c_void_p, C.c_int64]
for with_obs in (True, False):
    env = G.BatchedGobblet(n, "cuda:0", auto_reset=True, seed=0, with_observation=with_obs)
    env.rollout(64)
    buf = env.trajectory_buffers(T)
    for _ in range(6):
        env.collect(T, out=buf, refresh=False)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); env.collect(T, out=buf, refresh=False); b.record()
    torch.cuda.synchronize()
    us = a.elapsed_time(b) * 1e3
    ntiles = min(n // 64, 1 << 17)
    st = np.zeros((ntiles, 12), np.uint64)
    assert L.gbl_debug_stamps(st.ctypes.data, ntiles) == 0
    cyc = (st[:, 5] - st[:, 0]).astype(np.int64)
    ns = (st[:, 9] - st[:, 8]).astype(np.int64) * 10
    clock = cyc.sum() / ns.sum()
    by = ((178 if with_obs else 61) + 2) * T * n
    print(f"{'FULL' if with_obs else 'MASK_ONLY'} T={T} boards {n}: launch {us:.1f} us = {us / T:.2f} us/ply; wavefront life "
          f"{ns.mean() / 1e3:.1f} us; shader clock {clock:.2f} GHz; written {by / us / 1e6:.2f} TB/s = "
          f"{by / 256 / (us * 1e-6 * clock * 1e9):.2f} B/clk/CU")
