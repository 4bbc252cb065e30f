#!/usr/bin/env python3
"""Diagnostic (-DGBL_STAMPS build): where does a workgroup of k_greedy<4> spend its life?  Per tile (wave 0):
owner phase (loads, planes, depth-1 walk, pair lists), pooled cheap evaluations, deferred exact evaluations,
replay + outputs; plus the launch profile from the chip-wide 100 MHz clock."""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import gobblet_rl_amd as G  # noqa: E402

if os.environ.get("GOBBLET_HIP_LIB"):  # an experiment's own build of the library (scripts/build_variant.sh)
    G._native.use_library(os.environ["GOBBLET_HIP_LIB"])

boards = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
nt = int(sys.argv[2]) if len(sys.argv) > 2 else (4 if 32768 < boards <= 131072 else 1)  # tiles per workgroup of the shape in use
nat, L = G._native, G._native.lib()
env = G.BatchedGobblet(boards, "cuda:0", auto_reset=True, seed=0)
env.rollout(64)
act = torch.empty(boards, dtype=torch.int32, device="cuda:0")
cm = torch.empty((boards, 54), dtype=torch.int8, device="cuda:0")
fb = torch.empty(boards, dtype=torch.int8, device="cuda:0")
for _ in range(3):
    nat.check(L.gbl_greedy(env.squares.data_ptr(), env.to_move.data_ptr(), None, None, 2, act.data_ptr(), cm.data_ptr(),
                           fb.data_ptr(), boards, None))
torch.cuda.synchronize()
ntiles = min(boards // 64, 1 << 17)  # (stamps are indexed by the TILE of the flushing thread: blocks of NT tiles leave gaps)
buf = np.zeros((ntiles, 12), np.uint64)
L.gbl_debug_stamps.argtypes = [C.c_void_p, C.c_int64]
assert L.gbl_debug_stamps(buf.ctypes.data, ntiles) == 0
buf = buf[::nt]  # (a block's stamps are flushed by its first owner, under its tile's index)
ntiles = len(buf)
t = buf[:, :6].astype(np.int64)
d = np.diff(t, axis=1)
names = ["owner: loads, planes, depth-1 walk, pair lists", "pooled cheap evaluations (4 waves)", "deferred exact evaluations",
         "replay + outputs", "drain"]
print(f"boards {boards}, tiles {ntiles}; shader cycles per phase (wave 0 of each tile)")
for i, nm in enumerate(names):
    print(f"  {nm:48s} mean {d[:, i].mean():8.1f}  p50 {np.median(d[:, i]):8.1f}  p95 {np.percentile(d[:, i], 95):8.1f}")
life = t[:, 5] - t[:, 0]
print(f"  {'workgroup lifetime':48s} mean {life.mean():8.1f}  p50 {np.median(life):8.1f}  p95 {np.percentile(life, 95):8.1f}")
rt0, rt1 = buf[:, 8].astype(np.int64), buf[:, 9].astype(np.int64)
base = rt0.min()
st, en = (rt0 - base) * 10, (rt1 - base) * 10
print(f"  starts (ns after the first): p10 {np.percentile(st, 10):.0f} p50 {np.percentile(st, 50):.0f} p90 {np.percentile(st, 90):.0f} last {st.max()}")
print(f"  ends   (ns after the first start): first {en.min()} p10 {np.percentile(en, 10):.0f} p50 {np.percentile(en, 50):.0f} p90 {np.percentile(en, 90):.0f} last {en.max()}")
print(f"  lifetime ns mean {(en - st).mean():.0f} -> shader clock {life.mean() / (en - st).mean():.2f} GHz")
xcc = (buf[:, 7] & 0xF).astype(np.int64)
for x in sorted(set(xcc.tolist())):
    m = xcc == x
    print(f"    xcc {x}: {int(m.sum()):5d} tiles  start {st[m].min():5d}..{st[m].max():5d} ns  eval phase mean {d[m, 1].mean():8.0f} cycles  "
          f"lifetime mean {life[m].mean():8.0f}  last end {en[m].max():6d} ns")
cu = ((buf[:, 6] >> 8) & 0xF).astype(np.int64) | (((buf[:, 6] >> 12) & 0xF).astype(np.int64) << 4) | (xcc << 8)
ids, inv = np.unique(cu, return_inverse=True)
per = np.bincount(inv)
print(f"  CUs used {len(ids)}; tiles per CU min {per.min()} max {per.max()} (histogram {np.bincount(per).tolist()})")
worst = np.array([life[inv == i].max() for i in range(len(ids))])
print(f"  slowest tile per CU: p50 {np.median(worst):.0f} p95 {np.percentile(worst, 95):.0f} max {worst.max()} cycles; by tiles per CU: "
      + ", ".join(f"{k}: {worst[per == k].mean():.0f}" for k in sorted(set(per.tolist()))))
