#!/usr/bin/env python3
"""bench.py -- env-steps/s of the batched Gobblet hot path on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--boards B] [--mode step|rollout]

One "step" = one lockstep ply of the benchmark pipeline over this rank's shard of boards:
    mode step    (default): gbl_sample (masked-uniform action from the mask buffer) + gbl_step
                 (fused raw_env.step + observe: state, mask, obs, winner, reward, done written) --
                 two launches per ply, every ply's mask and observation materialised in HBM.
    mode rollout: gbl_rollout with every_ply=1 -- the same per-ply outputs, K plies in ONE launch
                 (state stays in registers between plies).
Workload: 2^20 boards per GPU, all reset, 64 warm-up plies of masked-random
play with auto-reset (stationary mix of game phases), then K timed plies; synthetic data, RNG
keyed (seed=0, global board id, ply) so results do not depend on the number of GPUs.  Boards shard
by contiguous global index, one shard per rank, no collective on the step path ("weak" scaling:
the per-GPU shard is fixed).  For N>1 launch with torchrun (the driver does), one rank per GPU.

Prints ONE JSON line on rank 0 (see the task's bench contract) including
    roofline     -- dominant kernel (k_step / k_rollout): algorithmic 234 B per env-step
                     x boards per launch / mean launch duration from HIP events
                    recorded on the launch stream inside the timed region, vs 8 TB/s HBM peak
    cpu_baseline -- the CPU oracle (a C port of the reference algorithm; kind "port") doing the
                    same pipeline on the host cores, on a bounded sample (rank 0, N=1 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

ALGO_BYTES_FULL = 234      # reads 33 + writes 201 per env-step
HBM_PEAK_GBPS = 8000.0     # MI355X_MICROARCH.md: 8.0 TB/s spec


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=64)
    ap.add_argument("--boards", type=int, default=1 << 20, help="boards per GPU")
    ap.add_argument("--mode", choices=["step", "rollout"], default="step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="target CPU time of the baseline sample")
    return ap.parse_args()


def cpu_baseline(boards, warmup, target_s):
    """The oracle timed on this host: same pipeline (sample from mask, fused step writing mask+obs,
    auto-reset), same workload (stationary mix after `warmup` plies), all host cores."""
    import numpy as np

    import oracle
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    n = min(boards, 1 << 18)
    s, tm, dn = oracle.batch_reset(n)
    oracle.batch_rollout(s, tm, dn, 0, 0, 0, warmup, threads=cores, want_obs=False, want_mask=False)
    mask = oracle.batch_legal_mask(s, tm)
    a = np.zeros(n, np.int32); w = np.zeros(n, np.int8); r = np.zeros((n, 2), np.int8)
    obs = np.zeros((n, 3, 3, 13), np.int8)
    t0 = time.perf_counter()
    oracle.batch_sample_step(s, tm, dn, a, w, r, mask, obs, 0, 0, warmup, threads=cores)
    t1 = time.perf_counter() - t0
    plies = max(2, min(256, int(target_s / max(t1, 1e-6))))
    t0 = time.perf_counter()
    for k in range(plies):
        oracle.batch_sample_step(s, tm, dn, a, w, r, mask, obs, 0, 0, warmup + 1 + k, threads=cores)
    dt = time.perf_counter() - t0
    # single-thread rate on a smaller slice
    n1 = min(n, 1 << 15)
    s1, tm1, dn1 = s[:n1].copy(), tm[:n1].copy(), dn[:n1].copy()
    m1 = mask[:n1].copy()
    t0 = time.perf_counter()
    for k in range(4):
        oracle.batch_sample_step(s1, tm1, dn1, a[:n1].copy(), w[:n1].copy(), r[:n1].copy(), m1, obs[:n1].copy(), 0, 0,
                                 999 + k, threads=1)
    d1 = time.perf_counter() - t0
    return {"value": n * plies / dt, "unit": "env-steps/s", "cores": cores, "kind": "port",
            "sample": f"{n} boards x {plies} plies (sample+step+mask+obs, auto-reset), C oracle, {cores} threads",
            "value_1core": n1 * 4 / d1}


def main():
    args = parse()
    import torch

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus N>1 must be launched with torch.distributed.run (one rank per GPU)")
        args.gpus = world
    if not torch.cuda.is_available():
        sys.exit("bench.py needs an MI355X (there is no CPU fallback for the product path)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)

    import gobblet_rl_amd as G

    boards = args.boards
    env = G.BatchedGobblet(boards, dev, illegal_mode="noop", auto_reset=True, seed=0, env_base=rank * boards)
    K, W = args.steps, args.warmup

    def one_ply():
        env.sample_actions()
        env.step(env.actions)

    # warm-up plies (untimed): decorrelate game phases, warm caches / code objects
    if args.mode == "step":
        for _ in range(W):
            one_ply()
    else:
        if W:
            env.rollout(W, every_ply=True)
    torch.cuda.synchronize(dev)

    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
          for _ in range(K if args.mode == "step" else 1)]
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    if args.mode == "step":
        lib, nat = G._native.lib(), G._native
        stream = nat.current_stream(dev)
        p = dict(sq=env.squares.data_ptr(), tm=env.to_move.data_ptr(), dn=env.done.data_ptr(),
                 ac=env.actions.data_ptr(), wi=env.winner.data_ptr(), rw=env.rewards.data_ptr(),
                 mk=env.action_mask.data_ptr(), ob=env.observation.data_ptr())
        for k in range(K):
            lib.gbl_sample(p["mk"], p["ac"], boards, env.seed, env.env_base, env.ply, stream)
            ev[k][0].record()
            rc = lib.gbl_step(p["sq"], p["tm"], p["dn"], p["ac"], p["wi"], p["rw"], p["mk"], p["ob"], boards, 0, 1,
                              stream)
            ev[k][1].record()
            env.ply += 1
        nat.check(rc, "gbl_step")
    else:
        ev[0][0].record()
        env.rollout(K, every_ply=True)
        ev[0][1].record()
    torch.cuda.synchronize(dev)
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize(dev)
    elapsed = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    kernel_ms = [a.elapsed_time(b) for a, b in ev]
    launches = len(kernel_ms)
    mean_kernel_s = sum(kernel_ms) / launches / 1e3
    units_per_launch = boards * (1 if args.mode == "step" else K)
    achieved = ALGO_BYTES_FULL * units_per_launch / mean_kernel_s / 1e9

    if rank == 0:
        total_steps = boards * K * world
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(tpath):  # HBM bytes per launch from a committed rocprofv3 --pmc run of this command
            try:
                tj = json.load(open(tpath))
                key = f"{args.mode}:{boards}"
                traffic = tj.get(key, {}).get("hbm_bytes_per_launch")
            except Exception:  # noqa: BLE001
                traffic = None
        out = {
            "metric": "env-steps/sec at 2^20 parallel boards per GPU, masked-random play, bit-exact mask/winner/obs",
            "value": total_steps / elapsed,
            "unit": "env-steps/s",
            "n_gpus": world,
            "steps": K,
            "warmup": W,
            "ms_per_step": elapsed / K * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "int8",
            "data": "synthetic",
            "config": {"workload": f"{boards} boards per GPU x {world} GPU(s), masked-random actions, auto-reset, "
                                   f"FULL outputs (state+mask+obs+winner+reward+done) every ply",
                       "boards_per_gpu": boards, "total_boards": boards * world, "mode": args.mode,
                       "launches_per_step": 2 if args.mode == "step" else round(1.0 / K, 6),
                       "sharding": f"contiguous board ranges, {world} shard(s), no collective on the step path"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic,
                         "kernel": "k_step<mask,obs>" if args.mode == "step" else "k_rollout<every_ply>",
                         "algorithmic_bytes_per_env_step": ALGO_BYTES_FULL,
                         "env_steps_per_launch": units_per_launch,
                         "mean_launch_us": mean_kernel_s * 1e6, "launches_timed": launches},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(boards, W, args.cpu_seconds)
        c = env.counters.cpu().tolist()
        out["config"]["games_finished_rank0"] = int(c[1])
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
