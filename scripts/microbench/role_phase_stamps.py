#!/usr/bin/env python3
"""Diagnostic (-DGBL_STAMPS -DGBL_AB_COLLECT_CFG build, build/lib_stamps.so): where do the cycles of a ply go in each role
wavefront of the role kernel (k_collect_small)?  Shader cycles per ply and phase, means over the workgroups of one launch:
    python scripts/microbench/role_phase_stamps.py BOARDS CFG [T]"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import gobblet_rl_amd as G  # noqa: E402

if os.environ.get("GOBBLET_HIP_LIB"):  # an experiment's own build of the library (scripts/build_variant.sh)
    G._native.use_library(os.environ["GOBBLET_HIP_LIB"])

n, cfg = int(sys.argv[1]), int(sys.argv[2])
T = int(sys.argv[3]) if len(sys.argv) > 3 else 32
nat = G._native
L = C.CDLL(os.path.abspath(os.environ.get("AB_LIB", "build/lib_stamps.so")))
for name in ("gbl_collect", "gbl_collect_variant"):
    res, args = nat.SIGNATURES[name]
    getattr(L, name).restype, getattr(L, name).argtypes = res, args
L.gbl_ab_collect_cfg.argtypes = [C.c_int]
L.gbl_debug_wave_stamps.argtypes = [C.c_void_p]
env = G.BatchedGobblet(n, "cuda:0", auto_reset=True, seed=0)
env.rollout(64)
buf = env.trajectory_buffers(T, placement="any")
f = buf["_full"]
L.gbl_ab_collect_cfg(cfg)
for i in range(4):
    rc = L.gbl_collect(env.squares.data_ptr(), env.to_move.data_ptr(), env.done.data_ptr(), f["actions"].data_ptr(), f["winner"].data_ptr(),
                       f["rewards"].data_ptr(), f["done"].data_ptr(), f["to_move"].data_ptr(), f["action_mask"].data_ptr(),
                       f["observation"].data_ptr(), n, buf["_ply_stride"], buf["_tile_stride"], 0, 0, 64 + i * T, None, T, 0, None, None, None)
    assert rc == 0
torch.cuda.synchronize()
st = np.zeros((1024, 16, 12), np.uint64)
assert L.gbl_debug_wave_stamps(st.ctypes.data) == 0
t = st.astype(np.float64)
used = t[:, 0, :8].sum(axis=1) > 0
t = t[used]
names = ["loop", "pick", "word+move+winner", "reset", "stores+scalars", "obs image", "legal mask", "mask image"]
print(f"boards {n} cfg {cfg} (variant {L.gbl_collect_variant(n, T, 1, 1)}), {T} plies per launch, {len(t)} workgroups: shader cycles per ply")
print("wave  " + "  ".join(f"{x:>16s}" for x in names) + "     total")
for w in range(16):
    m = t[:, w, :8].mean(axis=0) / T
    if m.sum() == 0:
        continue
    print(f"{w:4d}  " + "  ".join(f"{x:16.0f}" for x in m) + f"  {m.sum():8.0f}")
