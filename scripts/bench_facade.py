#!/usr/bin/env python3
"""Config 1 of BASELINE.json on the build's single-env facade (SURVEY.md 8d iii): the AEC loop of the
reference's examples/example_basic.py:50-67 over ``gobblet_v1.env()``, masked-uniform actions, timed
as env-steps/s.  One board per launch, so this measures launch + host round-trip latency, not the
GPU; the reference's own figure for the same loop is 297 steps/s/core (build container).

    python scripts/bench_facade.py --games 200
"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gobblet_rl_amd as G  # noqa: E402

if os.environ.get("GOBBLET_HIP_LIB"):  # an experiment's own build of the library (scripts/build_variant.sh)
    G._native.use_library(os.environ["GOBBLET_HIP_LIB"])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--games", type=int, default=200)
    ap.add_argument("--device", default="cuda:0", help='"cpu": the host flavour of the ABI (gbl_cpu_*), BASELINE config 1 as written')
    args = ap.parse_args()
    rng = np.random.default_rng(0)
    env = G.gobblet_v1.env(device=args.device)
    steps = 0
    wins = {"player_1": 0, "player_2": 0}
    t0 = None
    for game in range(args.games + 5):
        if game == 5:  # first games warm the runtime up
            t0 = time.perf_counter()
            steps = 0
        env.reset()
        for agent in env.agent_iter():
            observation, reward, termination, truncation, info = env.last()
            if termination or truncation:
                if reward > 0 and game >= 5:
                    wins[agent] += 1
                env.step(None)
            else:
                legal = np.flatnonzero(observation["action_mask"])
                env.step(int(legal[rng.integers(len(legal))]))
                steps += 1
    dt = time.perf_counter() - t0
    print(f"facade AEC loop on {args.device}: {args.games} games, {steps} env-steps in {dt:.2f} s = {steps / dt:.0f} env-steps/s "
          f"(mean game {steps / args.games:.1f} plies; player_1 won {wins['player_1']}, player_2 won {wins['player_2']})")


if __name__ == "__main__":
    main()
