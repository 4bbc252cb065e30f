#!/usr/bin/env python3
"""Per-ply time of the fused ply (gbl_rollout, plies=1) over a list of batch sizes, in ONE process:

    python scripts/sweep_sizes.py [--sizes 4096,131072,262144,1048576] [--modes full,mask] [--plies 300] [--reps 5]

For every (size, mode): 64 warm-up plies, then K plies captured as one hipGraph (device-resident ply index, so
every replay draws fresh plies) replayed `reps` times between HIP events.  Prints one JSON line per case:
us_per_ply (median and best replay / K, kernel boundaries included) and the implied fraction of the 8 TB/s
HBM peak at 234 (FULL) / 117 (MASK_ONLY) algorithmic bytes per env-step.  Run it under
`rocprofv3 --kernel-trace` and feed the database to `scripts/rocpd_summary.py bygrid` for kernel-only durations
per size.  GOBBLET_HIP_LIB selects a differently built library (A/B runs)."""
import argparse
import json
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gobblet_rl_amd as G  # noqa: E402

if os.environ.get("GOBBLET_HIP_LIB"):  # an experiment's own build of the library (scripts/build_variant.sh)
    G._native.use_library(os.environ["GOBBLET_HIP_LIB"])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sizes", default="4096,131072,262144,1048576")
    ap.add_argument("--modes", default="full,mask")
    ap.add_argument("--plies", type=int, default=300)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--traj", type=int, default=32, help="plies per launch of the trajectory modes (traj, trajmask)")
    ap.add_argument("--tag", default=os.environ.get("GOBBLET_HIP_LIB", "default"))
    args = ap.parse_args()
    for mode in args.modes.split(","):
        for n in (int(s) for s in args.sizes.split(",")):
            traj = mode.startswith("traj") or mode.startswith("tile")
            env = G.BatchedGobblet(n, "cuda:0", auto_reset=True, seed=0, with_observation=mode in ("full", "traj", "tile"))
            for _ in range(64):
                env.rollout(1)
            env.device_ply()
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            if traj:  # gbl_collect: args.traj plies per launch, every ply materialised in its trajectory slot
                T = args.traj
                launches = max(1, args.plies // T)  # (every launch reuses the same T slots)
                buf = env.trajectory_buffers(T, layout="tile" if mode.startswith("tile") else "time")
                env.collect(T, out=buf, refresh=False)
                torch.cuda.synchronize()
                with torch.cuda.graph(g):
                    for _ in range(launches):
                        env.collect(T, out=buf, refresh=False)
                    env.advance_ply()
                plies = launches * T
            else:
                plies = args.plies
                with torch.cuda.graph(g):
                    for _ in range(plies):
                        env.rollout(1)
                    env.advance_ply()
            g.replay()  # untimed: first launch of the instantiated graph
            torch.cuda.synchronize()
            us = []
            for _ in range(args.reps):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                g.replay()
                b.record()
                torch.cuda.synchronize()
                us.append(a.elapsed_time(b) * 1e3 / plies)
            # algorithmic bytes per env-step: SURVEY.md 8d (234 FULL / 117 MASK_ONLY); a trajectory launch reads and
            # writes the state once per T plies and writes the action: 178 + 55 / T (61 + 55 / T without observation)
            bytes_per = {"full": 234, "mask": 117, "traj": 178 + 57 / args.traj, "trajmask": 61 + 57 / args.traj,
                         "tile": 178 + 57 / args.traj, "tilemask": 61 + 57 / args.traj}[mode]
            med = statistics.median(us)
            print(json.dumps({"tag": args.tag + (f":T{args.traj}" if traj else ""), "mode": mode, "boards": n, "us_per_ply": round(med, 3),
                              "best_us": round(min(us), 3), "frac_of_8TBps": round(bytes_per * n / med / 8e6, 4),
                              "env_steps_per_s": n / med * 1e6}), flush=True)
            del g, env
            if traj:
                del buf


if __name__ == "__main__":
    main()
