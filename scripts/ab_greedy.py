#!/usr/bin/env python3
"""In-process A/B of gbl_greedy (depth 2) between differently built libraries (scripts/build_variant.sh): every library
is dlopen'ed, launched in turn on the same positions, and must give identical decisions.
    python scripts/ab_greedy.py BOARDS lib1.so lib2.so ..."""
import ctypes as C
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gobblet_rl_amd as G  # noqa: E402

if os.environ.get("GOBBLET_HIP_LIB"):  # an experiment's own build of the library (scripts/build_variant.sh)
    G._native.use_library(os.environ["GOBBLET_HIP_LIB"])

n = int(sys.argv[1])
paths = sys.argv[2:]
nat = G._native
dev = torch.device("cuda:0")
libs = []
for p in paths:
    L = C.CDLL(os.path.abspath(p))
    res, args = nat.SIGNATURES["gbl_greedy"]
    L.gbl_greedy.restype, L.gbl_greedy.argtypes = res, args
    libs.append(L)
env = G.BatchedGobblet(n, dev, auto_reset=True, seed=0)
env.rollout(64)
outs = [(torch.empty(n, dtype=torch.int32, device=dev), torch.empty((n, 54), dtype=torch.int8, device=dev),
         torch.empty(n, dtype=torch.int8, device=dev)) for _ in libs]
iters = 20


def run(i):
    a, c, f = outs[i]
    rc = libs[i].gbl_greedy(env.squares.data_ptr(), env.to_move.data_ptr(), None, None, 2, a.data_ptr(), c.data_ptr(), f.data_ptr(),
                            n, nat.current_stream(dev))
    assert rc == 0


for i in range(len(libs)):
    run(i)
torch.cuda.synchronize()
for i in range(1, len(libs)):
    if os.environ.get("AB_ALLOW_DIFF"):  # (timing experiments with an inexact variant)
        print(paths[i], "boards with another decision:", int((outs[0][0] != outs[i][0]).sum()), flush=True)
    else:
        assert all(torch.equal(x, y) for x, y in zip(outs[0], outs[i])), paths[i]
# `iters` launches per library as one hipGraph (AB_EAGER=1: eager launches, as rounds 2-4 timed it -- a floor build's kernel is
# shorter than the host's launch rate of ~7 us, which eager timing then reports instead)
graphs = []
for i in range(len(libs)):
    if os.environ.get("AB_EAGER"):
        graphs.append(None)
        continue
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(iters):
            run(i)
    g.replay()
    graphs.append(g)
torch.cuda.synchronize()
res = [[] for _ in libs]
for rnd in range(7):
    for i in range(len(libs)):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        run(i)
        a.record()
        if graphs[i] is None:
            for _ in range(iters):
                run(i)
        else:
            graphs[i].replay()
        b.record()
        torch.cuda.synchronize()
        res[i].append(a.elapsed_time(b) * 1e3 / iters)
for p, r in zip(paths, res):
    print(f"{os.path.basename(p):24s} boards {n}: median {statistics.median(r):7.2f} us   " + " ".join(f"{x:6.2f}" for x in r), flush=True)
