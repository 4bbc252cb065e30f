#!/usr/bin/env python3
"""Diagnostic: where does a wavefront of the fused kernel spend its life?  Needs the library built
with -DGBL_STAMPS (GOBBLET_HIP_LIB=.../lib_stamps.so); reads the per-wave s_memtime stamps of ONE
launch.  Shares, not absolute speed (the stamped build is not the shipped one)."""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import gobblet_rl_amd as G  # noqa: E402

if os.environ.get("GOBBLET_HIP_LIB"):  # an experiment's own build of the library (scripts/build_variant.sh)
    G._native.use_library(os.environ["GOBBLET_HIP_LIB"])

boards = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
env = G.BatchedGobblet(boards, "cuda:0", auto_reset=True)
for _ in range(64):
    env.rollout(1)
torch.cuda.synchronize()
env.rollout(1)
torch.cuda.synchronize()
ntiles = min(boards // 64, 1 << 17)
buf = np.zeros((ntiles, 12), np.uint64)
lib = G._native.lib()
lib.gbl_debug_stamps.argtypes = [C.c_void_p, C.c_int64]
assert lib.gbl_debug_stamps(buf.ctypes.data, ntiles) == 0
t = buf[:, :6].astype(np.int64)
names = ["launch->start", "load (tile + scalars arrive)", "compute (sample+step)", "stage + issue stores", "scalar stores",
         "drain (stores acknowledged)"]
print(f"boards {boards}, waves {ntiles}; s_memtime ticks (100 MHz on gfx9: 10 ns)")
d = np.diff(t, axis=1)
for i in range(5):
    print(f"  {names[i + 1]:34s} mean {d[:, i].mean():8.1f}  p50 {np.median(d[:, i]):8.1f}  p95 {np.percentile(d[:, i], 95):8.1f}")
life = t[:, 5] - t[:, 0]
print(f"  wave lifetime                      mean {life.mean():8.1f}  p50 {np.median(life):8.1f}  p95 {np.percentile(life, 95):8.1f}")
# s_memtime counters are per XCD/SE and not synchronised; s_memrealtime (100 MHz) is chip-wide: launch profile
rt0, rt1 = buf[:, 8].astype(np.int64), buf[:, 9].astype(np.int64)
base = rt0.min()
st, en = (rt0 - base) * 10, (rt1 - base) * 10  # ns
print(f"  wave starts (ns after the first): p10 {np.percentile(st, 10):.0f}  p50 {np.percentile(st, 50):.0f}  "
      f"p90 {np.percentile(st, 90):.0f}  last {st.max()}")
print(f"  wave ends   (ns after the first start): first {en.min()}  p10 {np.percentile(en, 10):.0f}  "
      f"p50 {np.percentile(en, 50):.0f}  p90 {np.percentile(en, 90):.0f}  last {en.max()}")
life_ns = en - st
print(f"  wave lifetime in ns: mean {life_ns.mean():.0f}  -> shader clock {life.mean() / life_ns.mean():.2f} GHz")
xcc = (buf[:, 7] & 0xF).astype(np.int64)
for x in sorted(set(xcc.tolist())):
    m = xcc == x
    print(f"    xcc {x}: {int(m.sum()):6d} waves  first start {st[m].min():6d}  last start {st[m].max():6d}  last end {en[m].max():6d}")
