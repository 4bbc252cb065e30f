#!/usr/bin/env python3
"""The headline pipeline with most of the device already allocated by the process (python scripts/placement_full_device.py
GIB): what the placement search does then (cap, probes, fallback) and what gbl_collect runs at, placed and as allocated."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import gobblet_rl_amd as G  # noqa: E402

if os.environ.get("GOBBLET_HIP_LIB"):  # an experiment's own build of the library (scripts/build_variant.sh)
    G._native.use_library(os.environ["GOBBLET_HIP_LIB"])

taken = int(sys.argv[1]) if len(sys.argv) > 1 else 200
dev = torch.device("cuda:0")
hog = torch.empty(taken << 30, dtype=torch.uint8, device=dev)
free = torch.cuda.mem_get_info()[0] / 2 ** 30
for pl in ("auto", "any"):
    rec = bench.short_run(G, torch, dev, 1 << 20, 320, 64, traj=8, placement=pl)
    print("%d GiB taken by the process, %.0f GiB free, placement %s: %.2f us per ply, roofline.frac %.3f, %s"
          % (taken, free, pl, rec["us_per_step"], rec["roofline"]["frac"], json.dumps(rec["trajectory_placement"])))
del hog
