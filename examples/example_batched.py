#!/usr/bin/env python3
"""The throughput path: N boards in lockstep on one MI355X.

    python examples/example_batched.py --boards 1048576 --plies 200 --policy random|greedy
"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gobblet_rl_amd as G  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--boards", type=int, default=1 << 20)
    ap.add_argument("--plies", type=int, default=200)
    ap.add_argument("--policy", default="random", choices=["random", "greedy"])
    args = ap.parse_args()
    env = G.BatchedGobblet(args.boards, "cuda:0", auto_reset=True, seed=0)
    pol = G.GreedyGobbletPolicy(depth=2) if args.policy == "greedy" else None
    p1 = torch.zeros((), dtype=torch.int64, device=env.device)
    p2 = torch.zeros((), dtype=torch.int64, device=env.device)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.plies):
        if pol is None:
            obs, rewards, done, winner = env.rollout(1)          # action sampled on device, fused with the step
        else:
            actions = pol.compute_actions_from_state(env.squares, env.to_move)
            obs, rewards, done, winner = env.step(actions)       # obs["observation"] (N,3,3,13), obs["action_mask"] (N,54)
        p1 += (winner == 1).sum()
        p2 += (winner == -1).sum()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    games = int(p1 + p2)
    print(f"{args.boards} boards x {args.plies} plies ({args.policy}): {args.boards * args.plies / dt:.3e} env-steps/s, "
          f"{games} games finished, player_1 won {int(p1) / max(1, games):.1%}")


if __name__ == "__main__":
    main()
