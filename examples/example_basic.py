#!/usr/bin/env python3
"""One masked-random game through the single-environment AEC surface -- the loop of the reference's
gobblet_rl/examples/example_basic.py:50-67, over the HIP engine (needs an MI355X).

    python examples/example_basic.py --render_mode text --seed 0
"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gobblet_rl_amd as G  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--render_mode", default="text", choices=["text", "text_full", "none"])
    ap.add_argument("--seed", type=int, default=None)
    args = ap.parse_args()
    if args.seed is not None:
        np.random.seed(args.seed)
    env = G.gobblet_v1.env(render_mode=None if args.render_mode == "none" else args.render_mode)
    env.reset()
    for agent in env.agent_iter():
        observation, reward, termination, truncation, info = env.last()
        if termination or truncation:
            print(f"Agent: ({agent}), Reward: {reward}, info: {info}")
            env.step(None)
        else:
            action_mask = observation["action_mask"]
            action = np.random.choice(np.arange(len(action_mask)), p=action_mask / np.sum(action_mask))
            env.step(int(action))


if __name__ == "__main__":
    main()
