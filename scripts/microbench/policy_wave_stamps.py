#!/usr/bin/env python3
"""Diagnostic (-DGBL_STAMPS build): the per-wavefront phase stamps of greedy_tile (see greedy_wave_stamps.py) inside
gbl_collect_policy's ply loop -- the stamps left are the LAST ply's.  The helpers reach barrier A of a ply as soon as the previous
decision's last barrier lets them go, the owners only after the previous decision's merge + replay AND the ply's own work (move,
winner, scalars, observation and mask rows): "at A" of the owners is that sum.
    GOBBLET_HIP_LIB=build/lib_stamps.so python scripts/microbench/policy_wave_stamps.py [boards] [plies]"""
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
T = int(sys.argv[2]) if len(sys.argv) > 2 else 16
L = G._native.lib()
env = G.BatchedGobblet(boards, "cuda:0", auto_reset=True, seed=0)
env.rollout(64)
buf = env.trajectory_buffers(T, placement="any", policy_outputs=True)
for _ in range(3):
    env.collect(T, out=buf, policies=("greedy", "greedy"), refresh=False)
torch.cuda.synchronize()
raw = np.zeros((1024, 16, 12), np.uint64)
L.gbl_debug_wave_stamps.argtypes = [C.c_void_p]
assert L.gbl_debug_wave_stamps(raw.ctypes.data) == 0
t = raw.astype(np.int64)
t = t[t[:, 0, 0] > 0]
waves = int((t[0, :, 0] > 0).sum())
base = t[:, :waves, 8].min(axis=1)[:, None, None]
rel = t[:, :waves, :] - base
print(f"boards {boards}, {T} plies per launch: {len(t)} blocks of {waves} wavefronts; last ply, mean shader cycles after the block's first wavefront reaches barrier A")
print("wave    at A  past A    at B    at C  past C   lists  past D  chunks done  past E")
for w in range(waves):
    m = rel[:, w].mean(axis=0)
    print(f"{w:4d} {m[8]:7.0f} {m[9]:7.0f} {m[10]:7.0f} {m[0]:7.0f} {m[1]:7.0f} {max(m[6], m[7]):7.0f} {m[2]:7.0f} {m[3]:10.0f} {m[5]:9.0f}")
