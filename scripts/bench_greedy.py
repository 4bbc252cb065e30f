#!/usr/bin/env python3
"""Config 5 of BASELINE.json: N boards x depth-2 greedy lookahead (GreedyGobbletPolicy.compute_action,
greedy_policy.py:38-221) in one batched call.  Positions are taken from the stationary masked-random
mix (64 warm-up plies, non-terminal by construction of auto-reset), empty action history.
Reports decisions/s, the measured legality tests / leaf evaluations per decision (counted by the
CPU oracle on a sample), and the oracle's own rate on the host."""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--boards", type=int, default=65536)
    ap.add_argument("--depth", type=int, default=2)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--check", type=int, default=4096, help="boards compared bit-for-bit with the oracle")
    args = ap.parse_args()
    import numpy as np
    import torch

    import gobblet_rl_amd as G
    import oracle
    nat = G._native
    dev = torch.device("cuda:0")
    n = args.boards
    env = G.BatchedGobblet(n, dev, auto_reset=True, seed=0)
    for _ in range(64):
        env.rollout(1)
    act = torch.empty(n, dtype=torch.int32, device=dev)
    cm = torch.empty((n, 54), dtype=torch.int8, device=dev)
    fb = torch.empty(n, dtype=torch.int8, device=dev)
    L = nat.lib()

    def run():
        nat.check(L.gbl_greedy(env.squares.data_ptr(), env.to_move.data_ptr(), None, None, args.depth, act.data_ptr(),
                               cm.data_ptr(), fb.data_ptr(), n, nat.current_stream(dev)), "gbl_greedy")

    run(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(args.iters):
        run()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / args.iters
    k = min(args.check, n)
    s, tm = env.squares[:k].cpu().numpy(), env.to_move[:k].cpu().numpy()
    oracle.greedy_work(reset=True)
    t0 = time.perf_counter()
    o = oracle.batch_greedy(s, tm, depth=args.depth)
    cpu_s = time.perf_counter() - t0
    tests, leaves = oracle.greedy_work()
    ok = (np.array_equal(act[:k].cpu().numpy(), o[0]) and np.array_equal(cm[:k].cpu().numpy(), o[1])
          and np.array_equal(fb[:k].cpu().numpy(), o[2]))
    print(json.dumps({"metric": "greedy depth-%d decisions/s" % args.depth, "boards": n, "ms_per_call": ms,
                      "decisions_per_s": n / (ms / 1e3), "parity_vs_oracle_on": k, "parity": bool(ok),
                      "cpu_oracle_decisions_per_s_1core": k / cpu_s,
                      "reference_work_per_decision": {"legality_tests": tests / k, "leaf_evaluations": leaves / k},
                      "leaf_evaluations_per_s_equivalent": n / (ms / 1e3) * leaves / k}))


if __name__ == "__main__":
    main()
