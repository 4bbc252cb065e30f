#!/usr/bin/env python3
"""A few EAGER launches of one pipeline, for rocprofv3 --pmc passes (counters are attributed per dispatch):
    python scripts/run_eager.py MODE BOARDS [LAUNCHES] [T]
MODE: full | mask (gbl_rollout, one ply per launch) | traj | trajmask (gbl_collect, T plies per launch) |
      step (gbl_step_ex: the next mover's draw fused into the step's launch, one launch per ply) |
      step2 (gbl_sample + gbl_step per ply) | greedy (gbl_greedy depth 2 on the stationary mix) |
      policy (gbl_collect_policy, greedy vs greedy, T plies per launch)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gobblet_rl_amd as G  # noqa: E402

if os.environ.get("GOBBLET_HIP_LIB"):  # an experiment's own build of the library (scripts/build_variant.sh)
    G._native.use_library(os.environ["GOBBLET_HIP_LIB"])

mode, n = sys.argv[1], int(sys.argv[2])
launches = int(sys.argv[3]) if len(sys.argv) > 3 else 10
T = int(sys.argv[4]) if len(sys.argv) > 4 else 32
env = G.BatchedGobblet(n, "cuda:0", auto_reset=True, seed=0, with_observation=mode in ("full", "traj", "greedy", "step", "step2", "policy"))
if mode in ("full", "mask"):  # warm up with another kernel, so that every k_rollout dispatch of the profile is a measured one
    for _ in range(4):
        env.collect(16)
else:
    env.rollout(64)
torch.cuda.synchronize()
if mode in ("traj", "trajmask"):
    buf = env.trajectory_buffers(T)
    for _ in range(launches):
        env.collect(T, out=buf)
elif mode == "policy":
    buf = env.trajectory_buffers(T, policy_outputs=True)
    for _ in range(launches):
        env.collect(T, out=buf, policies=("greedy", "greedy"), refresh=False)
elif mode == "step":
    acts = env.sample_actions().clone()
    for _ in range(launches):
        env.step(acts, next_actions=acts)
elif mode == "step2":
    for _ in range(launches):
        env.step(env.sample_actions())
elif mode == "greedy":
    nat, L = G._native, G._native.lib()
    act = torch.empty(n, dtype=torch.int32, device="cuda:0")
    cm = torch.empty((n, 54), dtype=torch.int8, device="cuda:0")
    fb = torch.empty(n, dtype=torch.int8, device="cuda:0")
    for _ in range(launches):
        nat.check(L.gbl_greedy(env.squares.data_ptr(), env.to_move.data_ptr(), None, None, 2, act.data_ptr(),
                               cm.data_ptr(), fb.data_ptr(), n, nat.current_stream(torch.device("cuda:0"))), "gbl_greedy")
else:
    for _ in range(launches):
        env.rollout(1)
torch.cuda.synchronize()
print("done", mode, n, launches)
