#!/usr/bin/env python3
"""Diagnostic (-DGBL_STAMPS build): per WAVEFRONT of a greedy workgroup, when does it pass the phases of greedy_tile?
Stamps (shader cycles after the block's first wavefront reaches barrier A): 8 at barrier A (boards loaded and published), 9 past
it, 10 at barrier B (depth-1 walk / root / nonplain set done), 0 at barrier C (plan), 1 past it, 6 / 7 pair and item lists
built, 2 past barrier D, 3 work chunks done, 5 past barrier E, 11 outputs issued.  Means over the blocks of one launch, one
row per wavefront index."""
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
buf = np.zeros((1024, 16, 12), np.uint64)
L.gbl_debug_wave_stamps.argtypes = [C.c_void_p]
assert L.gbl_debug_wave_stamps(buf.ctypes.data) == 0
t = buf.astype(np.int64)
used = t[:, 0, 0] > 0
t = t[used]
waves = int((t[0, :, 0] > 0).sum())
base = t[:, :waves, 8].min(axis=1)[:, None, None]
rel = t[:, :waves, :] - base
print(f"boards {boards}: {len(t)} blocks of {waves} wavefronts; mean shader cycles after the block's first wavefront reaches barrier A")
print("wave    at A  past A    at B    at C  past C   lists  past D  chunks done  past E   end   | B-phase  plan  lists  chunks  wait E  tail")
for w in range(waves):
    m = rel[:, w].mean(axis=0)
    end = m[11] if m[11] > 0 else float("nan")
    print(f"{w:4d} {m[8]:7.0f} {m[9]:7.0f} {m[10]:7.0f} {m[0]:7.0f} {m[1]:7.0f} {max(m[6], m[7]):7.0f} {m[2]:7.0f} {m[3]:10.0f} {m[5]:9.0f} {end:7.0f}"
          f"   | {m[10]-m[9]:7.0f} {m[0]-m[10]:5.0f} {max(m[6], m[7])-m[1]:6.0f} {m[3]-m[2]:7.0f} {m[5]-m[3]:7.0f} {end-m[5]:5.0f}")
