#!/usr/bin/env python3
"""Pure masked-random playouts: gbl_rollout with K plies per launch (boards stay in registers, only the
last ply's outputs are stored) -- the Monte-Carlo-playout use of SURVEY.md 8(f1).  No per-ply HBM
traffic, so this is integer-VALU bound; reports env-steps/s and games/s."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gobblet_rl_amd as G  # noqa: E402

if os.environ.get("GOBBLET_HIP_LIB"):  # an experiment's own build of the library (scripts/build_variant.sh)
    G._native.use_library(os.environ["GOBBLET_HIP_LIB"])

boards = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
K = int(sys.argv[2]) if len(sys.argv) > 2 else 64
env = G.BatchedGobblet(boards, "cuda:0", auto_reset=True, seed=0)
env.rollout(K, count=True)
torch.cuda.synchronize()
c0 = env.counters.clone()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
iters = 10
e0.record()
for _ in range(iters):
    env.rollout(K, count=True)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / iters
c = (env.counters - c0).tolist()
print(json.dumps({"metric": "masked-random playout plies/s (K plies per launch, outputs of the last ply only)",
                  "boards": boards, "plies_per_launch": K, "ms_per_launch": ms,
                  "env_steps_per_s": boards * K / (ms / 1e3), "games_per_s": c[1] / iters / (ms / 1e3),
                  "p1_win_rate": c[2] / max(1, c[1])}))
