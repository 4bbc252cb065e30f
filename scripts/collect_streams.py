#!/usr/bin/env python3
"""Which of gbl_collect's output streams costs what?  The same launch with subsets of the trajectory arrays (NULL
pointers switch an output off): all seven, rows only (mask + obs), obs only, mask only, scalars only."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gobblet_rl_amd as G  # noqa: E402

if os.environ.get("GOBBLET_HIP_LIB"):  # an experiment's own build of the library (scripts/build_variant.sh)
    G._native.use_library(os.environ["GOBBLET_HIP_LIB"])

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
T = int(sys.argv[2]) if len(sys.argv) > 2 else 8
nat, L = G._native, G._native.lib()
env = G.BatchedGobblet(n, "cuda:0", auto_reset=True, seed=0)
env.rollout(64)
buf = env.trajectory_buffers(T)
f = buf["_full"]
ctr = torch.zeros(1, dtype=torch.int32, device="cuda:0")
P = {k: v.data_ptr() for k, v in f.items()}


def run(keys, launches=32, reps=3):
    p = {k: (P[k] if k in keys else None) for k in P}

    def launch(off, stream):
        nat.check(L.gbl_collect(env.squares.data_ptr(), env.to_move.data_ptr(), env.done.data_ptr(), p["actions"], p["winner"],
                                p["rewards"], p["done"], p["to_move"], p["action_mask"], p["observation"], n,
                                buf["_ply_stride"], buf["_tile_stride"], 0, 0, off, ctr.data_ptr(), T, 0, None, None, stream))
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        s = nat.current_stream(torch.device("cuda:0"))
        for i in range(launches):
            launch(i * T, s)
        nat.check(L.gbl_counter_add(ctr.data_ptr(), launches * T, s))
    g.replay()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); g.replay(); b.record()
        torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b) * 1e3 / (launches * T))
    return best


sizes = {"actions": 4, "winner": 1, "rewards": 2, "done": 1, "to_move": 1, "action_mask": 54, "observation": 117}
scal = ("actions", "winner", "rewards", "done", "to_move")
for name, keys in (("all seven", tuple(sizes)), ("mask + obs", ("action_mask", "observation")), ("obs only", ("observation",)),
                   ("mask only", ("action_mask",)), ("scalars only", scal), ("mask + scalars", ("action_mask",) + scal),
                   ("obs + scalars", ("observation",) + scal), ("nothing", ())):
    us = run(keys)
    by = sum(sizes[k] for k in keys)
    print(f"{name:16s} {by:4d} B/step  {us:7.2f} us/ply  {by * n / us / 1e6:6.2f} TB/s", flush=True)
