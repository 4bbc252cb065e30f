#!/usr/bin/env python3
"""One ply per launch (the literal raw_env.step drop-in: a trainer with its policy outside the library): gbl_rollout_at(plies = 1)
on k_rollout against gbl_collect(plies = 1) on the collect kernels' forms writing the SAME environment tensors, in-process on a
library built with -DGBL_AB_COLLECT_CFG (build/lib_ab.so): 200 dependent launches as one hipGraph.
    python scripts/experiments/ab_oneply.py BOARDS cfg cfg ...     cfg: -2 = k_rollout, 0 = k_collect2 / k_collect, 3 = k_collect3, 100 LA + 10 KO + MERGE"""
import ctypes as C
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import gobblet_rl_amd as G  # noqa: E402

if os.environ.get("GOBBLET_HIP_LIB"):  # an experiment's own build of the library (scripts/build_variant.sh)
    G._native.use_library(os.environ["GOBBLET_HIP_LIB"])

n = int(sys.argv[1])
cfgs = [int(c) for c in sys.argv[2:]]
nat = G._native
L = C.CDLL(os.path.abspath(os.environ.get("AB_LIB", "build/lib_ab.so")))
for name in ("gbl_collect", "gbl_counter_add", "gbl_rollout_at"):
    res, args = nat.SIGNATURES[name]
    getattr(L, name).restype, getattr(L, name).argtypes = res, args
L.gbl_ab_collect_cfg.argtypes = [C.c_int]
env = G.BatchedGobblet(n, "cuda:0", auto_reset=True, seed=0)
env.rollout(64)
ctr = torch.zeros(1, dtype=torch.int32, device="cuda:0")
slot = -(-n // 128) * 128
K = 200
E = env


def launch(cfg, i, s):
    if cfg == -2:
        return L.gbl_rollout_at(E.squares.data_ptr(), E.to_move.data_ptr(), E.done.data_ptr(), E.actions.data_ptr(), E.winner.data_ptr(),
                                E.rewards.data_ptr(), E.action_mask.data_ptr(), E.observation.data_ptr(), n, 0, 0, i, ctr.data_ptr(), 1, 0,
                                None, None, s)
    L.gbl_ab_collect_cfg(cfg)
    return L.gbl_collect(E.squares.data_ptr(), E.to_move.data_ptr(), E.done.data_ptr(), E.actions.data_ptr(), E.winner.data_ptr(),
                         E.rewards.data_ptr(), None, None, E.action_mask.data_ptr(), E.observation.data_ptr(), n, slot, 64, 0, 0, i,
                         ctr.data_ptr(), 1, 0, None, None, s)


# every form leaves the same tensors as k_rollout (one launch each from one saved position)
saved = (E.squares.clone(), E.to_move.clone(), E.done.clone())
ref = None
for cfg in cfgs:
    E.squares.copy_(saved[0]); E.to_move.copy_(saved[1]); E.done.copy_(saved[2])
    assert launch(cfg, 7000, nat.current_stream(torch.device("cuda:0"))) == 0
    torch.cuda.synchronize()
    got = [t.clone() for t in (E.squares, E.to_move, E.done, E.actions, E.winner, E.rewards, E.action_mask, E.observation)]
    if ref is None:
        ref = got
    else:
        assert all(torch.equal(a, b) for a, b in zip(ref, got)), cfg
graphs = []
for cfg in cfgs:
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        s = nat.current_stream(torch.device("cuda:0"))
        for i in range(K):
            assert launch(cfg, i, s) == 0
        assert L.gbl_counter_add(ctr.data_ptr(), K, s) == 0
    g.replay()
    graphs.append(g)
torch.cuda.synchronize()
res = [[] for _ in cfgs]
for rnd in range(7):
    for i, g in enumerate(graphs):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); g.replay(); b.record()
        torch.cuda.synchronize()
        res[i].append(a.elapsed_time(b) * 1e3 / K)
print(f"boards {n}: {len(cfgs)} forms leave identical tensors")
for cfg, r in zip(cfgs, res):
    print(f"boards {n:7d} one ply per launch  cfg {cfg:4d}: median {statistics.median(r):6.3f} us/ply   min {min(r):6.3f}   "
          f"({n * 234 / statistics.median(r) / 8e6:.3f} of the HBM peak)", flush=True)
