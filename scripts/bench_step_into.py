#!/usr/bin/env python3
"""The collector loop of an EXTERNAL policy: per ply a policy launch (here: the library's sampler reading the previous
slot's mask) and BatchedGobblet.step_into() writing the ply into its trajectory slot -- two launches + three small
copies per ply, captured as one hipGraph of T plies.  With the trajectory arrays placed by the probe and as allocated.
usage: bench_step_into.py [boards] [T]"""
import json
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gobblet_rl_amd as G  # noqa: E402

if os.environ.get("GOBBLET_HIP_LIB"):  # an experiment's own build of the library (scripts/build_variant.sh)
    G._native.use_library(os.environ["GOBBLET_HIP_LIB"])

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
T = int(sys.argv[2]) if len(sys.argv) > 2 else 8
dev = torch.device("cuda:0")
res = {}
for placement in ("auto", "any", "auto", "any"):
    env = G.BatchedGobblet(n, dev, auto_reset=True, seed=0)
    env.rollout(64)
    env.device_ply()
    out = env.trajectory_buffers(T, placement=placement)
    acts = torch.zeros(n, dtype=torch.int32, device=dev)

    def plies():
        mask = env.action_mask
        for t in range(T):
            env.action_mask, keep = mask, env.action_mask   # the sampler reads the mask the last ply wrote
            env.sample_actions(out=acts)
            env.action_mask = keep
            _, mask = env.step_into(acts, out, t)
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
    m = statistics.median(us)
    res.setdefault(placement, []).append(round(m, 2))
    print(f"boards {n} T {T} placement {placement:4s}: {m:7.2f} us per ply = {n / m * 1e6:.3e} env-steps/s   {out['_placement']}", flush=True)
    keep_alive = (env, out) if placement == "any" else None
print(json.dumps({"boards": n, "T": T, "us_per_ply": res}))
