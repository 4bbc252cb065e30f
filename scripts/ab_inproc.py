#!/usr/bin/env python3
"""In-process A/B of gbl_collect between differently built libraries (scripts/build_variant.sh): every library is
dlopen'ed side by side and launched on the SAME buffers in turn (A B C A B C ...), so that buffer placement and the
box are the same for all of them -- the run-to-run scatter of the trajectory stream is larger than
most kernel variants' effect.

    python scripts/ab_inproc.py BOARDS T STREAMS lib1.so lib2.so ...      STREAMS: all | rows | obs | mask | none | scalars | maskscalars | obsscalars | ply | plymask
(ply / plymask: the one-ply pipeline instead, gbl_rollout_at with plies = 1, FULL / MASK_ONLY outputs; T is ignored)"""
import ctypes as C
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gobblet_rl_amd as G  # noqa: E402

if os.environ.get("GOBBLET_HIP_LIB"):  # an experiment's own build of the library (scripts/build_variant.sh)
    G._native.use_library(os.environ["GOBBLET_HIP_LIB"])

n, T, streams = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
paths = sys.argv[4:]
nat = G._native
libs = []
for p in paths:
    L = C.CDLL(os.path.abspath(p))
    for name in ("gbl_collect", "gbl_counter_add", "gbl_rollout_at"):
        res, args = nat.SIGNATURES[name]
        getattr(L, name).restype, getattr(L, name).argtypes = res, args
    libs.append(L)
env = G.BatchedGobblet(n, "cuda:0", auto_reset=True, seed=0)
env.rollout(64)
buf = env.trajectory_buffers(T)
f = buf["_full"]
one_ply = streams in ("ply", "plymask")
if one_ply:
    T = 1
scalars = ("actions", "winner", "rewards", "done", "to_move")
keys = {"all": tuple(f), "rows": ("action_mask", "observation"), "obs": ("observation",), "mask": ("action_mask",),
        "none": (), "scalars": scalars, "maskscalars": scalars + ("action_mask",), "obsscalars": scalars + ("observation",),
        "ply": (), "plymask": ()}[streams]
P = {k: (v.data_ptr() if k in keys else None) for k, v in f.items()}
ctr = torch.zeros(1, dtype=torch.int32, device="cuda:0")
launches = max(2, 256 // T)
graphs = []
for L in libs:
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        s = nat.current_stream(torch.device("cuda:0"))
        for i in range(launches):
            if one_ply:
                rc = L.gbl_rollout_at(env.squares.data_ptr(), env.to_move.data_ptr(), env.done.data_ptr(), env.actions.data_ptr(),
                                      env.winner.data_ptr(), env.rewards.data_ptr(), env.action_mask.data_ptr(),
                                      env.observation.data_ptr() if streams == "ply" else None, n, 0, 0, i, ctr.data_ptr(), 1, 0,
                                      None, None, s)
                assert rc == 0
                continue
            rc = L.gbl_collect(env.squares.data_ptr(), env.to_move.data_ptr(), env.done.data_ptr(), P["actions"], P["winner"],
                               P["rewards"], P["done"], P["to_move"], P["action_mask"], P["observation"], n, buf["_ply_stride"],
                               buf["_tile_stride"], 0, 0, i * T, ctr.data_ptr(), T, 0, None, None, s)
            assert rc == 0
        assert L.gbl_counter_add(ctr.data_ptr(), launches * T, s) == 0
    g.replay()
    graphs.append(g)
torch.cuda.synchronize()
res = [[] for _ in libs]
for rnd in range(7):
    for i, g in enumerate(graphs):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); g.replay(); b.record()
        torch.cuda.synchronize()
        res[i].append(a.elapsed_time(b) * 1e3 / (launches * T))
for p, r in zip(paths, res):
    print(f"{os.path.basename(p):28s} boards {n} T {T} {streams:5s}: median {statistics.median(r):7.2f} us/ply   "
          + " ".join(f"{x:6.2f}" for x in r), flush=True)
