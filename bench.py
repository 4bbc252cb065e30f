#!/usr/bin/env python3
"""bench.py -- env-steps/s of the batched Gobblet hot path on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--boards B] [--mode fused|step] [--graph 0|1]

One "step" = one lockstep ply of the benchmark pipeline over this rank's shard of boards:
    mode fused (default): gbl_rollout(plies=1) -- masked-uniform sampling + raw_env.step + observe
                 fused into ONE launch per ply; state, action, mask, obs, winner, reward, done of
                 every ply are materialised in HBM (a consumer can read them after each launch).
    mode step  : gbl_sample (action from the mask buffer) + gbl_step (externally supplied
                 actions: the drop-in form of raw_env.step + observe) -- two launches per ply.
By default the K timed plies are replayed as one hipGraph (captured and instantiated before the
timed region; every kernel node carries its own ply index); --graph 0 launches eagerly.
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
ALGO_BYTES_MASK_ONLY = 117  # same without the 117-byte observation
HBM_PEAK_GBPS = 8000.0     # MI355X_MICROARCH.md: 8.0 TB/s spec


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=64)
    ap.add_argument("--boards", type=int, default=1 << 20, help="boards per GPU (weak scaling, the default)")
    ap.add_argument("--total-boards", type=int, default=0,
                    help="fixed TOTAL board count split over the GPUs instead (strong scaling, e.g. 1048576)")
    ap.add_argument("--mode", choices=["step", "fused"], default="fused")
    ap.add_argument("--graph", type=int, default=1, help="1: replay the K timed plies as one hipGraph; 0: eager launches")
    ap.add_argument("--no-obs", action="store_true",
                    help="MASK_ONLY variant (BASELINE.md: 117 algorithmic bytes per env-step): no observation tensor")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"],
                    help="gloo + --share-device rehearses the multi-rank path on a one-GPU box")
    ap.add_argument("--share-device", action="store_true", help="all ranks use cuda:0 (rehearsal only)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="target CPU time of the baseline sample")
    return ap.parse_args()


def host_cores():
    """Cores this process may actually use: the affinity mask, cut down to the cgroup CPU quota."""
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            cores = min(cores, max(1, int(int(quota) / int(period) + 0.5)))
    except Exception:  # noqa: BLE001
        pass
    return max(1, min(cores, 64))


def cpu_baseline(boards, warmup, target_s):
    """The oracle timed on this host: same pipeline (sample from mask, fused step writing mask+obs,
    auto-reset), same workload (stationary mix after `warmup` plies), all host cores."""
    import numpy as np

    import oracle
    cores = host_cores()
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
    if args.share_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1 or os.environ.get("GBL_BENCH_FORCE_DIST"):  # (the env knob rehearses the RCCL path at world 1)
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)  # RCCL; used only for barriers + one MAX-reduce
        else:
            dist.init_process_group("gloo")

    import gobblet_rl_amd as G

    if args.total_boards:
        env_base, boards = G.shard_bounds(args.total_boards, world, rank)
    else:
        boards, env_base = args.boards, rank * args.boards
    env = G.BatchedGobblet(boards, dev, illegal_mode="noop", auto_reset=True, seed=0, env_base=env_base,
                           with_observation=not args.no_obs)
    algo_bytes = ALGO_BYTES_MASK_ONLY if args.no_obs else ALGO_BYTES_FULL
    K, W = args.steps, args.warmup
    lib, nat = G._native.lib(), G._native
    P = dict(sq=env.squares.data_ptr(), tm=env.to_move.data_ptr(), dn=env.done.data_ptr(),
             ac=env.actions.data_ptr(), wi=env.winner.data_ptr(), rw=env.rewards.data_ptr(),
             mk=env.action_mask.data_ptr(), ob=None if args.no_obs else env.observation.data_ptr())

    def enqueue_ply(ply, stream, ev=None):
        """One ply of the pipeline on `stream`; ev = (start, stop) events bracketing the dominant kernel."""
        if args.mode == "step":
            lib.gbl_sample(P["mk"], P["ac"], boards, env.seed, env.env_base, ply, stream)
            if ev:
                ev[0].record()
            rc = lib.gbl_step(P["sq"], P["tm"], P["dn"], P["ac"], P["wi"], P["rw"], P["mk"], P["ob"], None, boards, 0, 1,
                              stream)
        else:
            if ev:
                ev[0].record()
            rc = lib.gbl_rollout(P["sq"], P["tm"], P["dn"], P["ac"], P["wi"], P["rw"], P["mk"], P["ob"], boards,
                                 env.seed, env.env_base, ply, 1, 0, None, None, stream)
        if ev:
            ev[1].record()
        return rc

    # warm-up plies (untimed): decorrelate game phases, warm caches / code objects
    for k in range(W):
        nat.check(enqueue_ply(k, nat.current_stream(dev)), "warm-up")
    torch.cuda.synchronize(dev)

    graph = None
    if args.graph:
        # the K timed plies as ONE hipGraph (every kernel node carries its own ply index);
        # capture + instantiate happen here, outside the timed region; the replay is timed
        graph = torch.cuda.CUDAGraph()
        # thread_local: calls made by other threads of this process (e.g. RCCL's watchdog polling its
        # events in the multi-GPU runs) must not invalidate the capture
        with torch.cuda.graph(graph, capture_error_mode="thread_local"):
            cs = nat.current_stream(dev)
            for k in range(K):
                nat.check(enqueue_ply(W + k, cs), "capture")
        torch.cuda.synchronize(dev)
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))]
    else:
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(K)]
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    if graph is not None:
        ev[0][0].record()
        graph.replay()
        ev[0][1].record()
    else:
        stream = nat.current_stream(dev)
        for k in range(K):
            rc = enqueue_ply(W + k, stream, ev[k])
        nat.check(rc, "timed plies")
    torch.cuda.synchronize(dev)
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize(dev)
    elapsed = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev if args.dist_backend == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    kernel_ms = [a.elapsed_time(b) for a, b in ev]
    if graph is not None:
        # events bracket the whole replay: K dominant-kernel launches (+ K sampler launches in step
        # mode, which are subtracted pro rata by their share of algorithmic bytes: 58 of 292)
        share = 1.0 if args.mode == "fused" else algo_bytes / (algo_bytes + 58.0)
        mean_kernel_s = kernel_ms[0] * share / K / 1e3
        launches = K
    else:
        launches = len(kernel_ms)
        mean_kernel_s = sum(kernel_ms) / launches / 1e3
    units_per_launch = boards
    achieved = algo_bytes * units_per_launch / mean_kernel_s / 1e9

    if rank == 0:
        all_boards = args.total_boards if args.total_boards else boards * world
        total_steps = all_boards * K
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(tpath):  # HBM bytes per launch from a committed rocprofv3 --pmc run of this command
            try:
                tj = json.load(open(tpath))
                key = f"{args.mode}{'-noobs' if args.no_obs else ''}:{boards}"
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
            "scaling": "strong" if args.total_boards else "weak",
            "vs_baseline": None,
            "dtype": "int8",
            "data": "synthetic",
            "config": {"workload": f"{boards} boards per GPU x {world} GPU(s), masked-random actions, auto-reset, "
                                   f"{'MASK_ONLY' if args.no_obs else 'FULL'} outputs (state+mask{'' if args.no_obs else '+obs'}+winner+reward+done) every ply",
                       "boards_per_gpu": boards, "total_boards": all_boards, "mode": args.mode,
                       "launches_per_step": 2 if args.mode == "step" else 1,
                       "sharding": f"contiguous board ranges, {world} shard(s), no collective on the step path"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic,
                         "kernel": ("k_step" if args.mode == "step" else "k_rollout (plies=1)")
                                   + ("<mask>" if args.no_obs else "<mask,obs>"),
                         "algorithmic_bytes_per_env_step": algo_bytes,
                         "env_steps_per_launch": units_per_launch,
                         "mean_launch_us": mean_kernel_s * 1e6, "launches_timed": launches,
                         "timing": ("HIP events around the graph replay / K (includes kernel boundaries)"
                                    if graph is not None else "HIP event pair around every launch")},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(boards, W, args.cpu_seconds)
        out["config"]["launch"] = "hipGraph replay of K plies" if graph is not None else "eager"
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
