#!/usr/bin/env python3
"""In-process A/B of the role kernel's forms (k_collect_small<LA, KO, MERGE>) and k_collect2 / k_collect on ONE library built
with -DGBL_AB_COLLECT_CFG (scripts/build_variant.sh ab -DGBL_AB_COLLECT_CFG): gbl_ab_collect_cfg(cfg) picks the form at run
time, every form is captured into a hipGraph of its own on the SAME buffers and the graphs are replayed in turn.

    python scripts/ab_cfg.py BOARDS T STREAMS cfg cfg ...      cfg = 100 LA + 10 KO + MERGE; 0 = k_collect2 / k_collect; -1 = the library's choice
    STREAMS: all | mask | none | scalars (which trajectory arrays are passed)"""
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
cfgs = [int(c) for c in sys.argv[4:]]
nat = G._native
L = C.CDLL(os.path.abspath(os.environ.get("AB_LIB", "build/lib_ab.so")))
for name in ("gbl_collect", "gbl_counter_add", "gbl_collect_variant"):
    res, args = nat.SIGNATURES[name]
    getattr(L, name).restype, getattr(L, name).argtypes = res, args
L.gbl_ab_collect_cfg.argtypes = [C.c_int]
env = G.BatchedGobblet(n, "cuda:0", auto_reset=True, seed=0, with_observation=streams != "mask")
env.rollout(64)
# (AB_PLACEMENT=auto: the arrays placed across HBM's memory classes as bench.py's are -- from ~49 152 boards x 32 plies the
#  trajectory is an HBM stream and an unplaced pair costs every form alike 15-20 %)
buf = env.trajectory_buffers(T, placement=os.environ.get("AB_PLACEMENT", "any" if n < (1 << 19) else "auto"))
print("placement:", buf["_placement"].get("ratio"), buf["_placement"].get("ended", buf["_placement"].get("why")), flush=True)
f = buf["_full"]
scalars = ("actions", "winner", "rewards", "done", "to_move")
keys = {"all": tuple(f), "mask": scalars + ("action_mask",), "none": (), "scalars": scalars}[streams]
P = {k: (f[k].data_ptr() if k in keys and k in f else None) for k in ("actions", "winner", "rewards", "done", "to_move", "action_mask", "observation")}
ctr = torch.zeros(1, dtype=torch.int32, device="cuda:0")
launches = max(2, int(os.environ.get("AB_PLIES", "256")) // T)  # (AB_PLIES: plies per graph replay; long replays show the clocks' ramp)
graphs, names = [], []
for cfg in cfgs:
    L.gbl_ab_collect_cfg(cfg)
    names.append("%4d -> variant %d" % (cfg, L.gbl_collect_variant(n, T, 1, 0 if streams == "mask" else 1)))
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        s = nat.current_stream(torch.device("cuda:0"))
        for i in range(launches):
            rc = L.gbl_collect(env.squares.data_ptr(), env.to_move.data_ptr(), env.done.data_ptr(), P["actions"], P["winner"],
                               P["rewards"], P["done"], P["to_move"], P["action_mask"], P["observation"], n, buf["_ply_stride"],
                               buf["_tile_stride"], 0, 0, i * T, ctr.data_ptr(), T, 0, None, None, s)
            assert rc == 0, rc
        assert L.gbl_counter_add(ctr.data_ptr(), launches * T, s) == 0
    g.replay()
    graphs.append(g)
torch.cuda.synchronize()
# every form leaves the same trajectory and the same boards as the first one of the list (one launch each from one saved position)
saved = (env.squares.clone(), env.to_move.clone(), env.done.clone())
ref = None
for cfg in cfgs:
    env.squares.copy_(saved[0]); env.to_move.copy_(saved[1]); env.done.copy_(saved[2])
    for k in keys:
        if k in f:
            f[k].fill_(99)
    L.gbl_ab_collect_cfg(cfg)
    rc = L.gbl_collect(env.squares.data_ptr(), env.to_move.data_ptr(), env.done.data_ptr(), P["actions"], P["winner"],
                       P["rewards"], P["done"], P["to_move"], P["action_mask"], P["observation"], n, buf["_ply_stride"],
                       buf["_tile_stride"], 0, 0, 5000, None, T, 0, None, None, nat.current_stream(torch.device("cuda:0")))
    assert rc == 0
    torch.cuda.synchronize()
    got = {k: f[k].clone() for k in keys if k in f}
    got["squares"], got["to_move_env"], got["done_env"] = env.squares.clone(), env.to_move.clone(), env.done.clone()
    if ref is None:
        ref = got
    else:
        bad = [k for k in ref if not torch.equal(ref[k], got[k])]
        assert not bad, (cfg, bad)
print(f"boards {n}: {len(cfgs)} forms leave identical trajectories and boards", flush=True)
res = [[] for _ in cfgs]
for rnd in range(7):
    for i, g in enumerate(graphs):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); g.replay(); b.record()
        torch.cuda.synchronize()
        res[i].append(a.elapsed_time(b) * 1e3 / (launches * T))
for nm, r in zip(names, res):
    print(f"boards {n:7d} T {T} {streams:5s} cfg {nm:24s}: median {statistics.median(r):6.3f} us/ply   min {min(r):6.3f}", flush=True)
