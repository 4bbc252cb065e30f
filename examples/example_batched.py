#!/usr/bin/env python3
"""The throughput path: N boards in lockstep on one MI355X.

    python examples/example_batched.py --boards 1048576 --plies 200 --policy random|greedy [--graph K] [--collect T]
    python examples/example_batched.py --boards 65536 --plies 40 --policy greedy --opponent random
        (a policy OUTSIDE the library against a masked-random opponent, one launch per ply: the step itself leaves the opponent's
         next draw and a status byte per action behind -- gbl_step_ex)
    python examples/example_batched.py --boards 65536 --plies 64 --policy greedy --opponent random --collect 16
        (whole games greedy vs random inside the launches: the decisions are taken on the device, gbl_collect_policy)
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
    ap.add_argument("--opponent", default=None, choices=["random", "greedy"],
                    help="with --collect: player_2's policy (default: the same as --policy)")
    ap.add_argument("--collect", type=int, default=0,
                    help="T plies per launch with every ply kept (gbl_collect; with a greedy side gbl_collect_policy: the "
                         "reference's greedy lookahead decides on the device, two random opening plies per game as in "
                         "tutorials/GreedyAgent/tutorial_greedy.py) -- what a rollout collector hands a trainer: trajectory "
                         "tensors (T, N, ...)")
    args = ap.parse_args()
    env = G.BatchedGobblet(args.boards, "cuda:0", auto_reset=True, seed=0, track_turn=True)
    if args.collect:
        sides = (args.policy, args.opponent or args.policy)
        device_policy = dict(policies=sides, opening_plies=2) if "greedy" in sides else {}
        T, launches = args.collect, max(1, args.plies // args.collect)
        buf = env.trajectory_buffers(T, policy_outputs=bool(device_policy))  # reused by every launch: a replay buffer's staging area
        env.collect(T, out=buf, **device_policy)     # warm-up
        wins = torch.zeros(3, dtype=torch.int64, device=env.device)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(launches):
            tr = env.collect(T, out=buf, refresh=False, **device_policy)
            # a consumer would now read tr["observation"] (T,N,3,3,13), tr["action_mask"] (T,N,54), tr["actions"],
            # tr["rewards"], tr["done"], tr["to_move"]; here: tally the results of the games that ended
            wins += torch.bincount((tr["winner"][tr["done"] != 0] + 1).long(), minlength=3)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        games = int(wins[0] + wins[2])
        print(f"{args.boards} boards x {launches * T} plies ({sides[0]} vs {sides[1]}, {T} plies per launch, every ply kept): "
              f"{args.boards * launches * T / dt:.3e} env-steps/s incl. the tally, {games} games finished, player_1 won "
              f"{int(wins[2]) / max(1, games):.1%}")
        return
    pol = G.GreedyGobbletPolicy(depth=2) if args.policy == "greedy" else None
    p1 = torch.zeros((), dtype=torch.int64, device=env.device)
    p2 = torch.zeros((), dtype=torch.int64, device=env.device)
    versus_random = pol is not None and args.opponent == "random" and not args.graph
    if versus_random:
        # the loop of the reference's trainers with a random opponent (MultiAgentPolicyManager([agent, RandomPolicy])): player_1's
        # action comes from the policy, player_2's is the masked-uniform draw that the PREVIOUS step left in `draws` -- the step
        # samples it from the mask it stores (next_actions=), so the opponent costs no launch; `status` says what became of every
        # action (0 = played; 1 = illegal; 3 = outside [0, 54), where the reference's env() asserts)
        draws = env.sample_actions().clone()
        status = torch.zeros(args.boards, dtype=torch.int8, device=env.device)
        bad = torch.zeros((), dtype=torch.int64, device=env.device)

    def one_ply():
        nonlocal p1, p2
        if versus_random:
            nonlocal bad
            mine = pol.compute_actions_from_state(env.squares, env.to_move)
            actions = torch.where(env.to_move == 0, mine.to(torch.int32), draws)
            obs, rewards, done, winner = env.step(actions, status=status, next_actions=draws)
            bad += (status != 0).sum()
        elif pol is None:
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
    if versus_random:
        assert int(bad) == 0, "a greedy choice or a masked-random draw was flagged illegal"
    print(f"{args.boards} boards x {args.plies} plies ({args.policy}{' vs random, one launch per ply' if versus_random else ''}): "
          f"{args.boards * args.plies / dt:.3e} env-steps/s, "
          f"{games} games finished, player_1 won {int(p1) / max(1, games):.1%}")


if __name__ == "__main__":
    main()
