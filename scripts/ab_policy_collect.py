#!/usr/bin/env python3
"""In-process A/B of gbl_collect_policy (greedy vs greedy, depth 2) between differently built libraries: every library is
dlopen'ed and launched in turn from the same start position; trajectories must be identical.
    python scripts/ab_policy_collect.py BOARDS T lib1.so lib2.so ..."""
import ctypes as C
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gobblet_rl_amd as G  # noqa: E402

if os.environ.get("GOBBLET_HIP_LIB"):  # an experiment's own build of the library (scripts/build_variant.sh)
    G._native.use_library(os.environ["GOBBLET_HIP_LIB"])

n, T = int(sys.argv[1]), int(sys.argv[2])
paths = sys.argv[3:]
nat = G._native
dev = torch.device("cuda:0")
libs = []
for p in paths:
    L = C.CDLL(os.path.abspath(p))
    res, args = nat.SIGNATURES["gbl_collect_policy"]
    L.gbl_collect_policy.restype, L.gbl_collect_policy.argtypes = res, args
    libs.append(L)
env = G.BatchedGobblet(n, dev, auto_reset=True, seed=0)
env.rollout(64)
start = env.state_dict()
buf = env.trajectory_buffers(T, policy_outputs=True)
f = buf["_full"]
env.reset_policy_history()


def run(i, ply):
    rc = libs[i].gbl_collect_policy(env.squares.data_ptr(), env.to_move.data_ptr(), env.done.data_ptr(), env.policy_hist.data_ptr(),
                                    f["actions"].data_ptr(), f["winner"].data_ptr(), f["rewards"].data_ptr(), f["done"].data_ptr(),
                                    f["to_move"].data_ptr(), f["action_mask"].data_ptr(), f["observation"].data_ptr(),
                                    f["chosen"].data_ptr(), f["how"].data_ptr(), None, n, buf["_ply_stride"], buf["_tile_stride"], 0, 0,
                                    ply, None, T, 2, 2, 0, 0, None, None, nat.current_stream(dev))
    assert rc == 0


ref = None
for i in range(len(libs)):
    env.load_state_dict(start); env.reset_policy_history()
    run(i, 64)
    torch.cuda.synchronize()
    got = (f["actions"].clone(), f["how"].clone(), env.squares.clone())
    if ref is None:
        ref = got
    assert all(torch.equal(x, y) for x, y in zip(ref, got)), paths[i]
res = [[] for _ in libs]
for rnd in range(5):
    for i in range(len(libs)):
        env.load_state_dict(start); env.reset_policy_history()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        run(i, 64)
        a.record()
        for k in range(6):
            run(i, 64 + T * (k + 1))
        b.record()
        torch.cuda.synchronize()
        res[i].append(a.elapsed_time(b) * 1e3 / 6 / T)
for p, r in zip(paths, res):
    print(f"{os.path.basename(p):24s} boards {n} T {T}: median {statistics.median(r):7.2f} us per ply   " + " ".join(f"{x:6.2f}" for x in r), flush=True)
