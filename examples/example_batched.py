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
    ap.add_argument("--graph", type=int, default=0,
                    help="capture this many plies in one hipGraph and replay it (launch latency off the critical path)")
    args = ap.parse_args()
    env = G.BatchedGobblet(args.boards, "cuda:0", auto_reset=True, seed=0)
    pol = G.GreedyGobbletPolicy(depth=2) if args.policy == "greedy" else None
    p1 = torch.zeros((), dtype=torch.int64, device=env.device)
    p2 = torch.zeros((), dtype=torch.int64, device=env.device)
    def one_ply():
        nonlocal p1, p2
        if pol is None:
            obs, rewards, done, winner = env.rollout(1)          # action sampled on device, fused with the step
        else:
            actions = pol.compute_actions_from_state(env.squares, env.to_move)
            obs, rewards, done, winner = env.step(actions)       # obs["observation"] (N,3,3,13), obs["action_mask"] (N,54)
        p1 += (winner == 1).sum()
        p2 += (winner == -1).sum()

    if args.graph:
        # the sampler's ply index (and the policy's call index) move to device memory, so that every replay of
        # the captured plies draws fresh random numbers
        env.device_ply()
        if pol is not None:
            pol.device_calls()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            one_ply(); env.advance_ply()                         # warm-up outside the capture
            if pol is not None:
                pol.advance_calls()
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, capture_error_mode="thread_local"):
                for _ in range(args.graph):
                    one_ply()
                env.advance_ply()
                if pol is not None:
                    pol.advance_calls()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(max(1, args.plies // args.graph)):
                graph.replay()
            torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        played = max(1, args.plies // args.graph) * args.graph
        games = int(p1 + p2)
        print(f"{args.boards} boards x {played} plies ({args.policy}, hipGraph of {args.graph} plies): "
              f"{args.boards * played / dt:.3e} env-steps/s, {games} games finished, player_1 won "
              f"{int(p1) / max(1, games):.1%}")
        return
    one_ply()                                                     # warm-up (module loads) outside the timing
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.plies):
        one_ply()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    games = int(p1 + p2)
    print(f"{args.boards} boards x {args.plies} plies ({args.policy}): {args.boards * args.plies / dt:.3e} env-steps/s, "
          f"{games} games finished, player_1 won {int(p1) / max(1, games):.1%}")


if __name__ == "__main__":
    main()
