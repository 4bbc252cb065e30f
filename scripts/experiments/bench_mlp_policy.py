#!/usr/bin/env python3
"""What an EXTERNAL policy's collector loop costs per ply when the policy is a real (small) network, not a stand-in: a two-layer
MLP in torch (117 -> H -> 54, bf16 GEMMs through hipBLASLt, masked argmax) decides for every board, `BatchedGobblet.step_into`
plays the ply into its trajectory slot (one launch), T plies captured as one hipGraph.  Reported beside it: the same loop with the
library's sampler in the policy's place, and the environment's launch alone -- i.e. how much of such a loop the one-ply launch
boundary is.  The loop of the reference's trainers (examples/example_tianshou_DQN.py: policy(obs, mask) ->
env.step -> buffer.add), N boards at a time.
usage: bench_mlp_policy.py [boards] [T] [hidden]"""
import json
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import gobblet_rl_amd as G  # noqa: E402

if os.environ.get("GOBBLET_HIP_LIB"):  # an experiment's own build of the library (scripts/build_variant.sh)
    G._native.use_library(os.environ["GOBBLET_HIP_LIB"])

n = int(sys.argv[1]) if len(sys.argv) > 1 else 131072
T = int(sys.argv[2]) if len(sys.argv) > 2 else 16
H = int(sys.argv[3]) if len(sys.argv) > 3 else 256
dev = torch.device("cuda:0")
torch.manual_seed(0)
w1 = (torch.randn(117, H, device=dev) * 0.1).to(torch.bfloat16)
w2 = (torch.randn(H, 54, device=dev) * 0.1).to(torch.bfloat16)
results = {}
for who in ("mlp", "sampler", "env only"):
    env = G.BatchedGobblet(n, dev, auto_reset=True, seed=0)
    env.rollout(64)
    env.device_ply()
    out = env.trajectory_buffers(T)
    acts = torch.zeros(n, dtype=torch.int32, device=dev)
    x = torch.empty((n, 117), dtype=torch.bfloat16, device=dev)

    def policy(obs, mask):
        if who == "mlp":
            x.copy_(obs.reshape(n, 117))                       # int8 -> bf16
            q = torch.relu(x @ w1) @ w2                        # (n, 54)
            q = q.masked_fill(mask == 0, float("-inf"))
            acts.copy_(q.argmax(dim=1))
        elif who == "sampler":
            env.action_mask, keep = mask, env.action_mask      # the sampler reads the mask the last ply wrote
            env.sample_actions(out=acts)
            env.action_mask = keep

    def plies():
        obs, mask = env.observation, env.action_mask
        for t in range(T):
            policy(obs, mask)
            obs, mask = env.step_into(acts, out, t)
        env.advance_ply()

    plies()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        plies()
    g.replay()
    torch.cuda.synchronize()
    us = []
    for _ in range(7):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); g.replay(); b.record()
        torch.cuda.synchronize()
        us.append(a.elapsed_time(b) * 1e3 / T)
    results[who] = round(statistics.median(us), 2)
    print(f"boards {n}, {T} plies per graph, hidden {H}, policy = {who:8s}: {results[who]:7.2f} us per ply = "
          f"{n / results[who] * 1e6:.3e} env-steps/s", flush=True)
print(json.dumps({"boards": n, "T": T, "hidden": H, "us_per_ply": results}))
