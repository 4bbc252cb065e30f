#!/usr/bin/env python3
"""bench.py -- env-steps/s of the batched Gobblet hot path on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--boards B | --boards-per-gpu B]
                    [--mode collect|fused|step|step2] [--traj T] [--graph 0|1] [--no-obs] [--no-configs] [--no-cpu-baseline]

One "step" = one lockstep ply of the benchmark pipeline over this rank's shard of boards -- masked-uniform
sampling + raw_env.step + observe of the next mover, with EVERY ply's action, mask, obs, winner, reward, done and
next mover materialised in HBM where a consumer can read them:
    mode collect (default): gbl_collect -- T plies (--traj; default by shard size, auto_traj) per launch, ply t writing slot t of
                 trajectory arrays [T][boards][...] (what a rollout collector hands a trainer); the boards stay in
                 LDS / registers between the plies of a launch, so the state crosses HBM once per T plies.
    mode fused : gbl_rollout(plies=1) -- ONE launch per ply, the ply's outputs overwrite the environment's
                 tensors (round 1's pipeline; 234 algorithmic bytes per env-step).
    mode step  : gbl_step_ex -- externally supplied actions (the drop-in form of raw_env.step + observe: the action array is READ
                 from HBM like any policy's output) and, in the same launch, the next mover's masked-uniform draw from the mask the
                 launch stores, written over the action array: ONE launch per ply (round 6; the sampler's launch and its 58 bytes
                 per board are gone).
    mode step2 : gbl_sample (action from the mask buffer) + gbl_step -- two launches per ply (rounds 1-5's `step`).
Workload (BASELINE.json metric / BASELINE.md C4): 2^20 boards IN TOTAL, sharded by contiguous global index
over the N GPUs (131 072 per GPU at N = 8; "strong" scaling), all reset, W warm-up plies of masked-random play
with auto-reset (stationary mix of game phases), then K timed plies; synthetic data, RNG keyed (seed=0, global
board id, ply) so results do not depend on N.  No collective on the step path: RCCL carries the rendezvous, two warm-up
barriers, one MAX-reduce of the elapsed time and one all-gather of per-rank numbers; the barrier on both sides of the timed
span is the node's own (LocalBarrier, --span-barrier).  --boards-per-gpu B fixes the per-GPU shard instead
("weak" scaling).  For N>1 launch with torchrun (the driver does), one rank per GPU.
By default K timed plies that take more than one launch are replayed as one hipGraph whose kernel nodes take the
ply index from a device-resident counter (gbl_rollout_at + gbl_counter_add); a timed run of ONE launch (the
driver's 20 plies) is launched eagerly, with the ply index by value.  Either way the K plies are played once UNTIMED through the same code path
first and the timed pass plays K fresh plies; --graph 0 launches eagerly always.

Rank 0 prints ONE compact JSON line (< 4 kB; the task's bench contract) as the LAST line of stdout:
    the contract keys, `config` (a workload string + scalars), and two flat objects
    roofline     -- dominant kernel: algorithmic bytes per launch / mean launch duration from HIP events on the
                    launch stream, vs 8 TB/s HBM peak.  Bytes per env-step: 234 for k_rollout / k_step (SURVEY.md
                    8d: reads 33 + writes 201); for k_collect the same per-ply outputs (mask 54 + obs 117 + action 4 +
                    winner 1 + done 1 + to_move 1 = 178) and the state only once per launch (57 B per board);
                    `traffic` = HBM bytes per launch from the committed PMC passes of the same launch shape
    cpu_baseline -- the CPU oracle (a C port of the reference algorithm; kind "port") doing the same pipeline
                    on the host cores, on a bounded sample (rank 0, N=1 only)
The FULL record goes to --configs-out (default gpurun_out/bench_configs.json, named in config.configs_file) and to stderr:
    configs      -- (N=1 only) the other sizes BASELINE.json names, each a short run with its own roofline:
                    4 096 and 262 144 boards, the 131 072-board shard of C4 at 8 GPUs, 2^22 boards (beyond the
                    Infinity Cache), MASK_ONLY at 2^20, and config 5 (65 536 boards x depth-2 greedy)
    detail       -- per-rank kernel times and trajectory placements; cpu_baseline's greedy extras.
"""
import argparse
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

ALGO_BYTES_FULL = 234      # reads 33 + writes 201 per env-step
ALGO_BYTES_MASK_ONLY = 117  # same without the 117-byte observation
# gbl_collect: per ply and board it writes mask 54 + obs 117 + action 4 + winner 1 + done 1 + to_move 1 (the same
# outputs SURVEY.md 8(d) counts); the state is read (27 + to_move 1) and written (27 + to_move 1 + done 1) once per launch
ALGO_BYTES_COLLECT_PLY, ALGO_BYTES_COLLECT_PLY_MASK_ONLY, ALGO_BYTES_COLLECT_LAUNCH = 178, 61, 57
HBM_PEAK_GBPS = 8000.0     # MI355X_MICROARCH.md: 8.0 TB/s spec
SIMDS, CLOCK_GHZ = 1024, 2.4  # 256 CUs x 4 SIMDs; max shader clock (MI355X_MICROARCH.md)
TOTAL_BOARDS = 1 << 20
COLLECT_KERNELS = {0: "k_collect", 1: "k_collect (plain stores)", 2: "k_collect2", 4: "k_collect3", 5: "k_collect5"}  # gbl_collect_variant()


def collect_kernel_name(variant):
    """The kernel a gbl_collect_variant() code stands for (>= 1000: the role kernel's form, GBL_COLLECT_ROLES(la, ko, merge))."""
    if variant >= 1000:
        la, ko, merge = (variant - 1000) // 100, (variant - 1000) // 10 % 10, variant % 10
        return f"k_collect_small<{la} lanes/board, {ko} obs waves{', merged' if merge else ''}>"
    return COLLECT_KERNELS.get(variant, "k_collect?")

# The sub-records of an N = 1 run: name -> (boards, timed plies, MASK_ONLY, mode).  tests/test_gpu_bench_kernels.py compares
# the kernel instantiation each one times with the oracle, at the same batch size and through the same entry point.
CONFIG_RECORDS = {
    # BASELINE.md's own recipe for the headline -- 64 warm-up plies (the run's W), >= 1 000 timed plies, 8 plies per launch as one
    # hipGraph -- whatever --steps the caller asked for: the driver's command times ONE 20-ply launch, this record rides in its line
    "headline_recipe_1048576": (1 << 20, 1000, False, "collect"),
    # (the latency-bound sizes replay 2 048 / 1 024 plies: a graph's first launch costs ~10 us, 5 % of a 256-ply replay at 4 096 boards
    #  -- profiles/r05/burst_length.txt: 0.737 us per ply at 256 plies per replay, 0.703 at 2 048, 0.687 at 16 384)
    "c2_4096": (4096, 2048, False, "collect"), "c_16384": (16384, 1024, False, "collect"), "c_32768": (32768, 1024, False, "collect"),
    "c_65536": (65536, 512, False, "collect"), "c3_262144": (262144, 128, False, "collect"),
    "c4_shard_131072": (131072, 256, False, "collect"), "large_4194304": (1 << 22, 64, False, "collect"),
    # the same two shards at a trainer-realistic 32 plies per launch beside the tuned launch lengths above (a launch
    # boundary costs 3-6 us, which 1 024-ply launches amortise and a collector that trains every 32 plies does not)
    "c2_4096_T32": (4096, 2048, False, "collect"), "c4_shard_131072_T32": (131072, 256, False, "collect"),
    "maskonly_1048576": (1 << 20, 64, True, "collect"),
    # round 1's pipeline, one launch per ply (gbl_rollout, 234 algorithmic bytes per env-step)
    "single_ply_1048576": (1 << 20, 200, False, "fused"), "single_ply_262144": (262144, 200, False, "fused"),
    "single_ply_131072": (131072, 200, False, "fused"), "single_ply_4096": (4096, 200, False, "fused"),
    "single_ply_maskonly_1048576": (1 << 20, 200, True, "fused"),
    "single_ply_large_4194304": (1 << 22, 40, False, "fused"),
    # externally supplied actions, the next mover's draw fused into the step's launch (gbl_step_ex): one launch per ply
    "step_pipeline_1048576": (1 << 20, 200, False, "step"), "step_pipeline_131072": (131072, 200, False, "step"),
    # ... and rounds 1-5's form of it: gbl_sample + gbl_step, two launches per ply
    "step_two_launch_1048576": (1 << 20, 200, False, "step2"),
}
# ... plus two records with their own drivers: c5_greedy_65536 (greedy_run) and greedy_collect_65536 (greedy_collect_run)
EXTRA_RECORDS = ("c5_greedy_65536", "greedy_collect_65536", "step_reply_131072", "step_reply_262144")
# (rounds 3-4 also carried two_stream_single_ply_*: two half batches on two streams inside one hipGraph, a negative result --
#  0.35-0.39 / 0.48-0.58 of the peak against 0.57 / 0.73 on one stream, profiles/r03 and r04/bench_default_full.json; dropped)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=64)
    ap.add_argument("--boards", "--total-boards", dest="boards", type=int, default=TOTAL_BOARDS,
                    help="TOTAL boards, split over the GPUs (strong scaling: BASELINE.md C4)")
    ap.add_argument("--boards-per-gpu", type=int, default=0, help="fixed per-GPU shard instead (weak scaling)")
    ap.add_argument("--mode", choices=["collect", "fused", "step", "step2"], default="collect")
    ap.add_argument("--traj", type=int, default=0,
                    help="plies per launch in mode collect; 0 = by shard size (auto_traj: 1024 for <= 8192 boards per GPU ... 8 "
                         "at 2^20, 4 from 2^22), never more than --steps")
    ap.add_argument("--graph", type=int, default=1, help="1: replay the K timed plies as one hipGraph; 0: eager launches")
    ap.add_argument("--no-obs", action="store_true",
                    help="MASK_ONLY variant (BASELINE.md: 117 algorithmic bytes per env-step): no observation tensor")
    ap.add_argument("--placement", choices=["auto", "spread", "any"], default="auto",
                    help="placement of the trajectory arrays (BatchedGobblet.trajectory_buffers): auto = observation and "
                         "mask arrays in different 96 GiB classes of HBM, found with a probe; any = as the allocator hands them out")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-configs", action="store_true", help="skip the sub-records of the other BASELINE configs")
    ap.add_argument("--configs-out", default="gpurun_out/bench_configs.json",
                    help="where the FULL record goes (the sub-records of the other configs, per-rank lists); relative to the "
                         "repo root; '' = stderr only.  stdout carries the compact contract line only")
    ap.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"],
                    help="gloo + --share-device rehearses the multi-rank path on a one-GPU box")
    ap.add_argument("--share-device", action="store_true", help="all ranks use cuda:0 (rehearsal only)")
    ap.add_argument("--span-barrier", default="local", choices=["local", "dist"],
                    help="the barrier on both sides of the timed span: local = generation counters of the node's ranks in shared "
                         "memory (one node is the contract; ~1 us); dist = torch.distributed's (RCCL: an all-reduce + a "
                         "synchronize per barrier, which at 8 ranks lasts about as long as the shard's 20 plies)")
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
    # config 5's baseline: depth-2 greedy decisions on 2048 positions of the same stationary mix, one core, with the
    # reference's work per decision counted (legality tests / leaf evaluations of greedy_policy.py:84-157)
    k = 2048
    oracle.greedy_work(reset=True)
    t0 = time.perf_counter()
    oracle.batch_greedy(s[:k].copy(), tm[:k].copy(), depth=2)
    dg = time.perf_counter() - t0
    tests, leaves = oracle.greedy_work()
    # ... and the build's own CPU TWIN: the host flavour of the ABI (gbl_cpu_collect: the device header compiled
    # for the host, include/gobblet_cpu.h), the same pipeline with every ply materialised, all cores and one
    twin = {}
    try:
        import torch  # noqa: F401

        import gobblet_rl_amd as G
        raw = G._native.cpu_raw()
        nt = min(n, 1 << 17)
        tenv = G.BatchedGobblet(nt, "cpu", auto_reset=True, seed=0)
        tenv.rollout(warmup)
        T = 8
        tbuf = tenv.trajectory_buffers(T, placement="any")
        try:
            raw.gbl_cpu_set_threads(cores)
            tenv.collect(T, out=tbuf, refresh=False)
            t0 = time.perf_counter()
            reps = 0
            while time.perf_counter() - t0 < max(1.0, target_s / 4):
                tenv.collect(T, out=tbuf, refresh=False)
                reps += 1
            dtw = time.perf_counter() - t0
            raw.gbl_cpu_set_threads(1)
            t0 = time.perf_counter()
            tenv.collect(T, out=tbuf, refresh=False)
            d1w = time.perf_counter() - t0
        finally:
            raw.gbl_cpu_set_threads(0)  # (the library's thread count is process-wide: later host-flavour calls get the default back)
        twin = {"twin_value": nt * T * reps / dtw, "twin_value_1core": nt * T / d1w,
                "twin_sample": f"{nt} boards x {T * reps} plies, gbl_cpu_collect (host flavour of the ABI), {cores} threads"}
    except Exception as e:  # noqa: BLE001  (no C++ compiler on the box: the port's numbers stand alone)
        twin = {"twin_value": None, "twin_error": str(e)[:200]}
    return {"value": n * plies / dt, "unit": "env-steps/s", "cores": cores, "kind": "port",
            "sample": f"{n} boards x {plies} plies (sample+step+mask+obs, auto-reset), C oracle, {cores} threads",
            "value_1core": n1 * 4 / d1, **twin,
            "greedy_depth2": {"decisions_per_s_1core": k / dg, "sample": f"{k} positions, 1 thread",
                              "legality_tests_per_decision": tests / k, "leaf_evaluations_per_decision": leaves / k}}


def auto_traj(boards, steps, requested=0, no_obs=False):
    """Plies per gbl_collect launch: the requested value; else a timed run of up to 32 plies is ONE launch (the driver's
    20 plies: one kernel, no boundary inside the timed region); else by shard size, as measured (profiles/r05/
    plies_per_launch.txt) -- a launch boundary costs 3-6 us (the grid drains, the next one loads its state and primes its
    generators), which small shards amortise over long launches (4 096 boards: 0.65 us per ply at 32 plies per launch, 0.57
    at 256, 0.55 at 1 024), while large shards want SHORT launches: with T slots of every array open at once, hundreds of MB
    apart, the write stream loses DRAM page locality (2^22 boards: 4 plies per launch 110.6 us per ply, 8 plies 126.3; profiles/
    r05/large_plies_per_launch.txt).  The trajectory arrays stay <= ~3 GiB; never more than the timed run."""
    if requested > 0:
        return max(1, min(requested, steps))
    if steps <= 32:
        return max(1, steps)
    for limit, plies in ((8192, 1024), (16384, 512), (65536, 256), (131072, 128), (262144, 64), (524288, 32), ((1 << 22) - 1, 8)):
        if boards <= limit:
            if no_obs and plies == 8:
                plies = 16  # MASK_ONLY slots are a third of FULL's: 2^20 boards 8 / 12 / 16 / 20 / 24 / 32 plies per launch:
                #             10.3-10.5 / 10.1 / 9.8-9.9 / 9.9-10.0 / 10.2-10.3 / 12.7 us per ply (plies_per_launch.txt)
            return min(plies, steps)
    return 4


def kernel_source_hash():
    """Identifies the kernel sources a committed profile was taken from (profiles/pmc_traffic.json records it)."""
    h = hashlib.sha256()
    for f in ("gobblet_hip.hip", "gobblet_device.h", "gobblet_knobs.h"):
        h.update(open(os.path.join(ROOT, "gobblet-rl_amd", "csrc", f), "rb").read())
    return h.hexdigest()[:16]


def committed_counter(key, field):
    """A per-launch counter value from profiles/pmc_traffic.json (rocprofv3 --pmc passes of this very command,
    committed with the hash of the kernel sources they were measured on), or (None, reason) when there is none
    or the kernels have changed since."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        rec = json.load(open(path)).get(key)
    except Exception:  # noqa: BLE001
        return None, "profiles/pmc_traffic.json missing"
    if not rec or field not in rec:
        return None, f"no committed counters for {key}"
    if rec.get("kernel_source_hash") != kernel_source_hash():
        return None, f"stale: {rec.get('source', 'profiles/pmc_traffic.json')} was measured on other kernel sources"
    return rec[field], f"{rec.get('source', 'profiles/pmc_traffic.json')} @ kernel sources {rec['kernel_source_hash']}"


class LocalBarrier:
    """A barrier of the ranks of ONE node (the contract: one process per GPU of one node): a generation counter per rank, a cache
    line apart, in a shared-memory block; arrive = publish the next generation, then wait until every rank has.  The contract
    brackets the timed span with barrier + synchronize; what a barrier costs is harness, not the path measured -- RCCL's is an
    all-reduce kernel plus a synchronize on every rank."""

    def __init__(self, dist, rank, world):
        self.rank, self.world, self.gen, self.dist, self.shm, self.slots = rank, world, 0, dist, None, None

    @classmethod
    def create(cls, dist, rank, world):
        """The barrier, or None when some rank could not map the block (no usable /dev/shm, ranks that do not share one): every
        rank learns the outcome, so that all of them fall back to torch.distributed's barrier together."""
        import numpy as np
        from multiprocessing import resource_tracker, shared_memory
        self = cls(dist, rank, world)
        name = [None]
        if rank == 0:
            try:
                self.shm = shared_memory.SharedMemory(create=True, size=64 * world)  # (zero-filled)
                name[0] = self.shm.name
            except OSError as e:
                print(f"bench.py: no shared-memory block for the span barrier ({e}): torch.distributed's instead", file=sys.stderr)
        dist.broadcast_object_list(name, src=0)
        ok = name[0] is not None
        if ok and rank != 0:
            try:
                self.shm = shared_memory.SharedMemory(name=name[0])
                resource_tracker.unregister(self.shm._name, "shared_memory")  # (rank 0 owns the block: it alone unlinks it)
            except OSError as e:
                print(f"bench.py: rank {rank} cannot map the span barrier's block ({e})", file=sys.stderr)
                ok = False
        oks = [None] * world
        dist.all_gather_object(oks, ok)  # (also: every rank is attached before anybody arrives)
        if not all(oks):
            self._release()
            return None
        self.slots = np.ndarray((world, 8), dtype=np.int64, buffer=self.shm.buf)
        return self

    def _release(self):
        self.slots = None
        if self.shm is not None:
            self.shm.close()
            if self.rank == 0:
                self.shm.unlink()
            self.shm = None

    def wait(self, timeout_s=120.0):
        self.gen += 1
        self.slots[self.rank, 0] = self.gen
        col, gen, deadline = self.slots[:, 0], self.gen, time.perf_counter() + timeout_s
        while int(col.min()) < gen:
            if time.perf_counter() > deadline:
                raise RuntimeError(f"bench.py: rank {self.rank} waited {timeout_s:.0f} s at the span barrier (generations {col.tolist()})")

    def close(self):
        self.dist.barrier()  # nobody is still spinning on the block
        self._release()


class Pipeline:
    """One shard of boards and its launch sequence, with the ply index in device memory."""

    def __init__(self, G, torch, boards, env_base, dev, no_obs=False, mode="collect", traj=32, placement="auto"):
        self.G, self.torch, self.nat, self.lib = G, torch, G._native, G._native.lib()
        self.boards, self.dev, self.mode, self.no_obs, self.T = boards, dev, mode, no_obs, max(1, int(traj))
        env = self.env = G.BatchedGobblet(boards, dev, illegal_mode="noop", auto_reset=True, seed=0, env_base=env_base,
                                          with_observation=not no_obs)
        self.P = dict(sq=env.squares.data_ptr(), tm=env.to_move.data_ptr(), dn=env.done.data_ptr(),
                      ac=env.actions.data_ptr(), wi=env.winner.data_ptr(), rw=env.rewards.data_ptr(),
                      mk=env.action_mask.data_ptr(), ob=None if no_obs else env.observation.data_ptr())
        self.ctr = torch.zeros(1, dtype=torch.int32, device=dev)  # plies played so far (keys the sampler)
        self.host_ply = 0      # ... and the host's copy of it: an eager launch can take the ply index BY VALUE
        self.owed = 0          # plies played by value that the device-resident index has not been told of yet (settle())
        self.traj = None
        if mode == "step":  # the action array holds the draw for ply 0; every launch leaves the next ply's behind
            self.nat.check(self.lib.gbl_sample_at(self.P["mk"], self.P["ac"], boards, env.seed, env.env_base, 0, None,
                                                  self.nat.current_stream(dev)), "gbl_sample_at")
        if mode == "collect":
            self.traj = env.trajectory_buffers(self.T, placement=placement, far=True)  # (the benchmark owns the device)
            f = self.traj["_full"]
            self.TP = dict(ac=f["actions"].data_ptr(), wi=f["winner"].data_ptr(), rw=f["rewards"].data_ptr(),
                           dn=f["done"].data_ptr(), tm=f["to_move"].data_ptr(), mk=f["action_mask"].data_ptr(),
                           ob=None if no_obs else f["observation"].data_ptr())

    def plan(self, k):
        """The launches that play k plies: [(ply offset, plies of the launch)]."""
        if self.mode != "collect":
            return [(i, 1) for i in range(k)]
        return [(i, min(self.T, k - i)) for i in range(0, k, self.T)]

    def launch_bytes(self, plies):
        """Algorithmic bytes of one launch of the dominant kernel (see the module docstring)."""
        if self.mode == "collect":
            per = ALGO_BYTES_COLLECT_PLY_MASK_ONLY if self.no_obs else ALGO_BYTES_COLLECT_PLY
            return (per * plies + ALGO_BYTES_COLLECT_LAUNCH) * self.boards
        return (ALGO_BYTES_MASK_ONLY if self.no_obs else ALGO_BYTES_FULL) * self.boards

    def enqueue(self, off, plies, stream, ev=None, by_value=False):
        """Plies (counter + off) .. (counter + off + plies - 1) on `stream`; ev = (start, stop) events bracketing
        the dominant kernel.  by_value (mode collect, eager launches): the ply index goes in as an argument instead of being read
        from device memory, so no launch has to advance it afterwards (played(): 7 us of a one-launch span,
        scripts/experiments/span_overhead.py)."""
        P, lib, env, n = self.P, self.lib, self.env, self.boards
        assert not by_value or (self.mode == "collect" and self.host_ply is not None)
        if self.mode == "step":
            if ev:
                ev[0].record()
            rc = lib.gbl_step_ex(P["sq"], P["tm"], P["dn"], P["ac"], P["wi"], P["rw"], P["mk"], P["ob"], None, None, None, None,
                                 None, P["ac"], env.seed, env.env_base, off + 1, self.ctr.data_ptr(), n, 0, 1, stream)
        elif self.mode == "step2":
            rc = lib.gbl_sample_at(P["mk"], P["ac"], n, env.seed, env.env_base, off, self.ctr.data_ptr(), stream)
            self.nat.check(rc, "gbl_sample_at")
            if ev:
                ev[0].record()
            rc = lib.gbl_step(P["sq"], P["tm"], P["dn"], P["ac"], P["wi"], P["rw"], P["mk"], P["ob"], None, n, 0, 1, stream)
        elif self.mode == "fused":
            if ev:
                ev[0].record()
            rc = lib.gbl_rollout_at(P["sq"], P["tm"], P["dn"], P["ac"], P["wi"], P["rw"], P["mk"], P["ob"], n, env.seed,
                                    env.env_base, off, self.ctr.data_ptr(), 1, 0, None, None, stream)
        else:
            if ev:
                ev[0].record()
            T = self.TP
            rc = lib.gbl_collect(P["sq"], P["tm"], P["dn"], T["ac"], T["wi"], T["rw"], T["dn"], T["tm"], T["mk"], T["ob"],
                                 n, self.traj["_ply_stride"], self.traj["_tile_stride"], env.seed, env.env_base,
                                 off + (self.host_ply if by_value else 0), None if by_value else self.ctr.data_ptr(), plies, 0,
                                 None, None, stream)
        self.nat.check(rc, "launch")
        if ev:
            ev[1].record()

    def advance(self, k, stream):
        self.nat.check(self.lib.gbl_counter_add(self.ctr.data_ptr(), k + self.owed, stream), "gbl_counter_add")
        if self.host_ply is not None:
            self.host_ply += k
        self.owed = 0

    def played(self, k):
        """k plies were launched with the ply index by value: the host's index moves on, the device's is settled later."""
        self.host_ply += k
        self.owed += k

    def settle(self, stream):
        if self.owed:
            self.advance(0, stream)

    def eager(self, k, events=None):
        s = self.nat.current_stream(self.dev)
        for i, (off, plies) in enumerate(self.plan(k)):
            self.enqueue(off, plies, s, events[i] if events else None)
        self.advance(k, s)

    def capture(self, k):
        torch = self.torch
        g = torch.cuda.CUDAGraph()
        # thread_local: calls made by other threads of this process (e.g. RCCL's watchdog polling its
        # events in the multi-GPU runs) must not invalidate the capture
        assert not self.owed
        self.host_ply = None  # (replays move the device-resident index behind the host's back: no by-value launches from here on)
        with torch.cuda.graph(g, capture_error_mode="thread_local"):
            cs = self.nat.current_stream(self.dev)
            for off, plies in self.plan(k):
                self.enqueue(off, plies, cs)
            self.advance(k, cs)
        return g

    def events(self, k):
        E = self.torch.cuda.Event
        return [(E(enable_timing=True), E(enable_timing=True)) for _ in range(k)]

    def kernel_roofline(self, kernel_s, plies_timed, launches, timing):
        """kernel_s: summed duration of the dominant kernel over `launches` launches that played `plies_timed` plies."""
        total_bytes = sum(self.launch_bytes(pl) for _, pl in self.plan(plies_timed))
        achieved = total_bytes / kernel_s / 1e9
        # which form of the kernel the library runs for this shape: asked of the library, not re-derived here
        variant = self.lib.gbl_collect_variant(self.boards, self.T, 1, 0 if self.no_obs else 1)
        collect = collect_kernel_name(variant)
        name = {"step": "k_step<EXT> (next draw fused)", "step2": "k_step", "fused": "k_rollout (plies=1)",
                "collect": f"{collect} ({self.T} plies per launch)"}[self.mode]
        # SURVEY.md 8(d)'s figure for the same env-steps: 234 (117 MASK_ONLY) bytes per env-step, state and action crossing HBM every
        # ply.  gbl_collect keeps the board in LDS for the T plies of a launch and WRITES the action instead of reading it, so it
        # moves fewer bytes than that and the fraction on SURVEY's bytes can exceed 1: it says how far the K-plies-per-launch design
        #  is ahead of a ply-per-launch pipeline running AT the HBM peak, it is not an achieved bandwidth.
        survey = ALGO_BYTES_MASK_ONLY if self.no_obs else ALGO_BYTES_FULL
        frac_survey = survey * self.boards * plies_timed / kernel_s / 1e9 / HBM_PEAK_GBPS
        accounting = ("SURVEY 8d, every ply: 234 B/board (117 MASK_ONLY); reward 2 B written, not counted" if self.mode != "collect" else
                      "state once per launch: (178*T+57) B/board (MASK_ONLY 61*T+57); reward 2 B/step written, not counted")
        return {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBPS, "traffic": None,
                "algorithmic_bytes_survey": survey, "frac_on_survey_bytes": frac_survey, "accounting": accounting,
                "kernel": name + ("<mask>" if self.no_obs else "<mask,obs>"),
                "algorithmic_bytes_per_env_step": total_bytes / (self.boards * plies_timed),
                "algorithmic_bytes_per_launch": total_bytes / launches,
                "env_steps_per_launch": self.boards * plies_timed / launches,
                "mean_launch_us": kernel_s / launches * 1e6, "launches_timed": launches, "timing": timing}


def traffic_key(mode, no_obs, boards, plies_per_launch):
    """The key of profiles/pmc_traffic.json for a pipeline (scripts/profile_round.sh takes the counters per key)."""
    return f"{mode}{'-noobs' if no_obs else ''}:{boards}" + (f":T{plies_per_launch}" if mode == "collect" else "")


def attach_traffic(roof, p, plies_per_launch):
    """roofline.traffic of pipeline p: measured HBM bytes per launch of its dominant kernel (committed PMC passes of the
    same launch shape), and their ratio to the algorithmic bytes of that launch; null (with the reason) when the
    committed counters were taken on other kernel sources or there are none for this shape."""
    key = traffic_key(p.mode, p.no_obs, p.boards, plies_per_launch)
    roof["traffic"], roof["traffic_source"] = committed_counter(key, "hbm_bytes_per_launch")
    if roof["traffic"] is not None:
        roof["traffic_plies_per_launch"] = plies_per_launch
        roof["traffic_algorithmic_bytes"] = p.launch_bytes(plies_per_launch)
        roof["traffic_over_algorithmic"] = roof["traffic"] / roof["traffic_algorithmic_bytes"]
    return roof


def short_run(G, torch, dev, boards, K, W, no_obs=False, mode="collect", traj=32, placement="auto"):
    """A sub-record: W warm-up plies, K plies as a hipGraph replayed once untimed, then timed between HIP events."""
    p = Pipeline(G, torch, boards, 0, dev, no_obs=no_obs, mode=mode, traj=traj, placement=placement)
    p.eager(W)
    g = p.capture(K)
    g.replay()
    torch.cuda.synchronize(dev)
    a, b = p.events(1)[0]
    a.record()
    g.replay()
    b.record()
    torch.cuda.synchronize(dev)
    s = a.elapsed_time(b) / 1e3
    launches = len(p.plan(K))
    rec = {"workload": f"{boards} boards x 1 GPU, masked-random actions, auto-reset, "
                       f"{'MASK_ONLY' if no_obs else 'FULL'} outputs every ply, mode {mode}, {K} plies "
                       f"({launches} launches) as one hipGraph",
           "value": boards * K / s, "unit": "env-steps/s", "us_per_step": s / K * 1e6,
           "roofline": attach_traffic(p.kernel_roofline(s, K, launches,
                                                        "HIP events around the graph replay (includes kernel boundaries)"),
                                      p, min(p.T, K) if mode == "collect" else 1)}
    if mode == "step2":
        rec["roofline"]["note"] = ("two kernels per ply (k_sample 58 B + k_step 234 B algorithmic per board): the fraction is "
                                   "the whole ply's time against k_step's 234 bytes")
    if mode == "step":
        rec["roofline"]["note"] = ("one kernel per ply: SURVEY 8(d)'s 234 bytes (the action is read from HBM) + the 4 bytes of the next "
                                   "action it writes, which the fraction does not count")
    if p.traj is not None:
        rec["trajectory_placement"] = p.traj["_placement"]
    del g, p
    return rec


def shard_span_run(G, torch, dev, boards, K, W, placement="auto"):
    """What ONE rank of an N-GPU run of the driver's command does, timed on this GPU: a shard of `boards` boards, W warm-up plies,
    the K plies rehearsed once, then the K plies inside a host span with a synchronize on both sides (the contract's span without
    the barrier, which one process cannot rehearse).  Returns (span seconds, kernel seconds by HIP events)."""
    T = auto_traj(boards, K)
    p = Pipeline(G, torch, boards, 0, dev, mode="collect", traj=T, placement=placement)
    p.eager(W)
    plan = p.plan(K)
    graph = p.capture(K) if len(plan) > 1 else None
    ev = p.events(1 if graph is not None else len(plan))
    stream = p.nat.current_stream(dev)

    def play():
        if graph is not None:
            ev[0][0].record(); graph.replay(); ev[0][1].record()
        else:
            for i, (off, plies) in enumerate(plan):
                p.enqueue(off, plies, stream, ev[i], by_value=True)
            p.played(K)
    play()
    torch.cuda.synchronize(dev)
    spans, kernels = [], []
    for _ in range(5):
        t0 = time.perf_counter()
        play()
        torch.cuda.synchronize(dev)
        spans.append(time.perf_counter() - t0)
        kernels.append(sum(a.elapsed_time(b) for a, b in ev) / 1e3)
    p.settle(stream)
    i = sorted(range(5), key=lambda j: spans[j])[2]  # the median pass
    return spans[i], kernels[i]


def scale_prediction(G, torch, dev, total, K, W, value_n1, placement="auto"):
    """Predicted 2 / 4 / 8-GPU lines of THIS command from one GPU (no multi-GPU node was ever available to the builder; a SCALE
    record can be checked against it): the boards shard without a collective, so an N-GPU run is N copies of the total / N-board
    shard's span side by side; what one GPU cannot show is the span barrier's skew between ranks (bench.py's is node-local shared
    memory, ~1 us) and host jitter across N processes -- the prediction is an upper bound by those."""
    out = {}
    for n in (2, 4, 8):
        span, kern = shard_span_run(G, torch, dev, total // n, K, W, placement)
        value = total * K / span
        out[str(n)] = {"boards_per_gpu": total // n, "span_us": span * 1e6, "kernel_us": kern * 1e6, "value_predicted": value,
                       "efficiency_predicted": value / (n * value_n1),
                       "efficiency_kernel_only": (total * K / kern) / (n * value_n1)}
    return out


def step_reply_run(G, torch, dev, boards, K, W):
    """Candidate (b): an external policy's ply and the masked-random reply in ONE launch (gbl_collect_from, 2 plies per
    launch, both materialised in trajectory slots).  The stand-in for the external policy is the library's sampler on
    the previous launch's last mask slot (its launch is part of the timed loop; the env-only time is taken separately with
    an event pair around every gbl_collect_from launch of an eager run)."""
    nat, L = G._native, G._native.lib()
    env = G.BatchedGobblet(boards, dev, auto_reset=True, seed=0)
    env.rollout(W)
    buf = env.trajectory_buffers(2)
    f = buf["_full"]
    ctr = torch.zeros(1, dtype=torch.int32, device=dev)
    env.refresh()
    f["action_mask"][1, :boards].copy_(env.action_mask)
    acts = torch.zeros(boards, dtype=torch.int32, device=dev)
    mask1 = f["action_mask"][1]

    def decision(off, stream, ev=None):
        nat.check(L.gbl_sample_at(mask1.data_ptr(), acts.data_ptr(), boards, 12345, 0, off, ctr.data_ptr(), stream), "gbl_sample_at")
        if ev:
            ev[0].record()
        nat.check(L.gbl_collect_from(env.squares.data_ptr(), env.to_move.data_ptr(), env.done.data_ptr(), acts.data_ptr(),
                                     f["actions"].data_ptr(), f["winner"].data_ptr(), f["rewards"].data_ptr(),
                                     f["done"].data_ptr(), f["to_move"].data_ptr(), f["action_mask"].data_ptr(),
                                     f["observation"].data_ptr(), boards, buf["_ply_stride"], buf["_tile_stride"], 0, 0, off,
                                     ctr.data_ptr(), 2, 0, None, None, stream), "gbl_collect_from")
        if ev:
            ev[1].record()
    D = K // 2
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, capture_error_mode="thread_local"):
        cs = nat.current_stream(dev)
        for i in range(D):
            decision(2 * i, cs)
        nat.check(L.gbl_counter_add(ctr.data_ptr(), 2 * D, cs), "gbl_counter_add")
    g.replay()
    torch.cuda.synchronize(dev)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); g.replay(); b.record()
    torch.cuda.synchronize(dev)
    sec = a.elapsed_time(b) / 1e3
    # the environment's share alone: the same launches without the stand-in's, replayed from a graph with a FIXED action
    # array (stale actions are mostly illegal -> no-ops; the kernel's work and traffic do not depend on that)
    g2 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g2, capture_error_mode="thread_local"):
        cs = nat.current_stream(dev)
        for i in range(D):
            nat.check(L.gbl_collect_from(env.squares.data_ptr(), env.to_move.data_ptr(), env.done.data_ptr(), acts.data_ptr(),
                                         f["actions"].data_ptr(), f["winner"].data_ptr(), f["rewards"].data_ptr(),
                                         f["done"].data_ptr(), f["to_move"].data_ptr(), f["action_mask"].data_ptr(),
                                         f["observation"].data_ptr(), boards, buf["_ply_stride"], buf["_tile_stride"], 0, 0,
                                         2 * i, ctr.data_ptr(), 2, 0, None, None, cs), "gbl_collect_from")
        nat.check(L.gbl_counter_add(ctr.data_ptr(), 2 * D, cs), "gbl_counter_add")
    g2.replay()
    torch.cuda.synchronize(dev)
    a.record(); g2.replay(); b.record()
    torch.cuda.synchronize(dev)
    kern = a.elapsed_time(b) / 1e3 / D
    evs = [None] * D
    per_launch = (2 * ALGO_BYTES_COLLECT_PLY + ALGO_BYTES_COLLECT_LAUNCH + 4) * boards
    variant = collect_kernel_name(L.gbl_collect_variant(boards, 2, 1, 1))
    return {"workload": f"{boards} boards x 1 GPU, an external policy's ply (stand-in: gbl_sample on the last mask slot) + the "
                        f"masked-random reply per launch (gbl_collect_from, 2 plies per launch), auto-reset, FULL outputs every ply, "
                        f"{2 * D} plies as one hipGraph",
            "value": boards * 2 * D / sec, "unit": "env-steps/s", "us_per_step": sec / (2 * D) * 1e6,
            "us_per_step_env_only": kern / 2 * 1e6,
            "roofline": {"bound": "hbm", "achieved": per_launch / kern / 1e9, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": per_launch / kern / 1e9 / HBM_PEAK_GBPS, "traffic": None,
                         "kernel": f"{variant} (2 plies per launch, first ply given)<mask,obs>",
                         "algorithmic_bytes_per_env_step": per_launch / boards / 2, "algorithmic_bytes_per_launch": per_launch,
                         "mean_launch_us": kern * 1e6, "launches_timed": len(evs),
                         "timing": "HIP events around a graph replay of the gbl_collect_from launches alone (fixed action array); "
                                   "value / us_per_step are the loop with the policy stand-in's launch"},
            "trajectory_placement": buf["_placement"]}


def greedy_run(G, torch, dev, boards=65536, iters=50):
    """BASELINE config 5: `boards` positions of the stationary masked-random mix x depth-2 greedy lookahead,
    one gbl_greedy launch per call.  Bound: integer VALU issue, not HBM (144 B of I/O per decision)."""
    nat, L = G._native, G._native.lib()
    env = G.BatchedGobblet(boards, dev, auto_reset=True, seed=0)
    env.rollout(64)
    act = torch.empty(boards, dtype=torch.int32, device=dev)
    cm = torch.empty((boards, 54), dtype=torch.int8, device=dev)
    fb = torch.empty(boards, dtype=torch.int8, device=dev)

    def run():
        nat.check(L.gbl_greedy(env.squares.data_ptr(), env.to_move.data_ptr(), None, None, 2, act.data_ptr(),
                               cm.data_ptr(), fb.data_ptr(), boards, nat.current_stream(dev)), "gbl_greedy")
    for _ in range(3):
        run()
    torch.cuda.synchronize(dev)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        run()
    b.record()
    torch.cuda.synchronize(dev)
    s = a.elapsed_time(b) / 1e3 / iters
    return {"workload": f"{boards} boards (stationary masked-random mix, both movers, empty history) x depth-2 greedy "
                        f"lookahead, one launch per call", "value": boards / s, "unit": "decisions/s",
            "us_per_step": s * 1e6,
            "roofline": valu_roofline(f"greedy:{boards}", "k_greedy (depth 2)", s, iters,
                                      "HIP events around back-to-back eager launches / count")}


VALU_CYCLES_PEAK, VALU_CYCLES_ONE_WAVE, VALU_CYCLES_MIX = 2.0, 4.0, 3.6
# SIMD cycles (of the nominal 2.4 GHz) per wave64 VALU instruction.  PEAK: a CDNA4 SIMD is 32 lanes wide and issues a wave64
# instruction over two cycles (MI355X_MICROARCH.md: 4 is one wavefront alone on its SIMD) -- the yardstick of `frac`.  MIX: what
# THIS code measures with every SIMD full -- scripts/microbench/reply_rate.hip times the greedy pair evaluation itself
# (profiles/r03/reply_rate.txt: 3.3-3.6); ONE_WAVE: rounds 1-3 quoted their fraction against it.  Both are named extras.


def valu_roofline(counter_key, kernel, launch_s, launches, timing):
    """Roofline of an integer-VALU-bound kernel: executed wave64 VALU instructions per launch (rocprofv3 SQ_INSTS_VALU of a
    committed profile of the same launch) against what 1024 SIMDs can issue in the launch's time at 2 cycles per instruction."""
    insts, src = committed_counter(counter_key, "SQ_INSTS_VALU")
    peak = SIMDS * CLOCK_GHZ * 1e9 / VALU_CYCLES_PEAK
    roof = {"bound": "valu", "unit": "wave64 VALU instructions/s", "peak": peak, "achieved": None, "frac": None,
            "frac_of_one_wave_per_simd_rate": None, "frac_of_measured_issue_rate": None, "kernel": kernel,
            "mean_launch_us": launch_s * 1e6, "launches_timed": launches, "timing": timing,
            "valu_instructions_per_launch": insts, "valu_instructions_source": src,
            "note": "frac = executed VALU instructions x 2 cycles / (1024 SIMDs x launch time x 2.4 GHz): the SIMD-32 issue peak; "
                    "frac_of_one_wave_per_simd_rate = the same with 4 cycles (rounds 1-3's yardstick); frac_of_measured_issue_rate = "
                    "with 3.6 cycles, the rate the pair evaluation itself issues at with every SIMD full "
                    "(profiles/r03/reply_rate.txt); HBM traffic is ~150-330 B per decision: irrelevant"}
    if insts:
        roof["achieved"] = insts / launch_s
        roof["frac"] = roof["achieved"] / peak
        roof["frac_of_one_wave_per_simd_rate"] = roof["achieved"] / (SIMDS * CLOCK_GHZ * 1e9 / VALU_CYCLES_ONE_WAVE)
        roof["frac_of_measured_issue_rate"] = roof["achieved"] / (SIMDS * CLOCK_GHZ * 1e9 / VALU_CYCLES_MIX)
    return roof


def greedy_collect_run(G, torch, dev, boards=65536, T=16, launches=8, policies=("greedy", "greedy")):
    """Whole games with the reference's greedy policy on both sides (tutorials/GreedyAgent/tutorial_greedy.py), T plies per
    launch with every ply materialised: gbl_collect_policy.  One env-step = one decision + raw_env.step + observe."""
    env = G.BatchedGobblet(boards, dev, auto_reset=True, seed=0)
    env.rollout(64)
    env.device_ply()
    buf = env.trajectory_buffers(T, policy_outputs=True)
    for _ in range(2):
        env.collect(T, out=buf, policies=policies, refresh=False)
        env.advance_ply()
    torch.cuda.synchronize(dev)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(launches):
        env.collect(T, out=buf, policies=policies, refresh=False)
        env.advance_ply()
    b.record()
    torch.cuda.synchronize(dev)
    s = a.elapsed_time(b) / 1e3 / launches
    greedy_share = float((buf["how"] != 0).float().mean())
    rec = {"workload": f"{boards} boards x 1 GPU, {policies[0]} vs {policies[1]} (device-side GreedyGobbletPolicy, depth 2, "
                       f"fallback draws and histories included), auto-reset, FULL outputs every ply, {T} plies per launch",
           "value": boards * T / s, "unit": "env-steps/s", "us_per_step": s / T * 1e6,
           "greedy_decisions_per_s": boards * T * greedy_share / s, "share_of_plies_decided_by_greedy": greedy_share,
           "roofline": valu_roofline(f"policy-collect:{boards}:T{T}", f"k_collect_policy ({T} plies per launch)", s, launches,
                                     "HIP events around back-to-back eager launches / count"),
           "trajectory_placement": buf["_placement"]}
    return rec


COMPACT_LIMIT = 4096  # bytes: the driver keeps the tail of stdout only; the contract line must fit it with room to spare
ROOFLINE_KEYS = ("bound", "achieved", "peak", "unit", "frac", "traffic", "traffic_over_algorithmic", "traffic_source", "kernel",
                 "algorithmic_bytes_per_env_step", "algorithmic_bytes_per_launch", "algorithmic_bytes_survey", "frac_on_survey_bytes",
                 "accounting", "mean_launch_us", "launches_timed", "timing")
# The sub-records every driver-parsed line carries as [value, us per step, fraction of the bound's peak]: the
# other BASELINE configs (C2, C3, C4's shard, C5), the SURVEY-8(d)-clean one-ply kernel (234 B per env-step) and the externally
# stepped pipeline.  The full records stay in the configs file.
COMPACT_CONFIGS = ("headline_recipe_1048576", "c2_4096", "c3_262144", "c4_shard_131072", "c5_greedy_65536", "single_ply_1048576", "step_pipeline_1048576",
                   "step_pipeline_131072", "greedy_collect_65536", "maskonly_1048576")


def sig(x, digits=4):
    """x rounded to `digits` significant digits (the compact line carries numbers, not noise)."""
    return None if x is None else float(f"{x:.{digits}g}")
CPU_BASELINE_KEYS = ("value", "unit", "cores", "kind", "sample", "value_1core", "twin_value", "twin_value_1core")


def contract_record(args, p, roof, total, boards, world, K, W, elapsed, local_elapsed, nlaunch, graphed, per_rank_us,
                    per_rank_placement, distributed, repeats_us):
    """The whole record of the headline run: the contract's keys, a `config` of one workload string and scalars (per-rank
    lists are kept under `detail`, which only the configs file and stderr carry), and the dominant kernel's roofline."""
    variant = "MASK_ONLY" if args.no_obs else "FULL"
    ratios = [pl["ratio"] for pl in per_rank_placement if pl and pl.get("ratio") is not None]
    reps = sorted(repeats_us)
    return {
        "metric": "env-steps/sec at 2^20 parallel boards, 1/2/4/8 MI355X; bit-exact mask/winner",
        # THE CONTRACT'S SPAN: the K plies of every rank (and the launch that moves the device-resident ply index on), bracketed
        # by a barrier + synchronize on both sides, MAX over ranks.  (Round 4 computed `value` from each rank's span WITHOUT the
        # trailing barrier -- its numbers are not comparable with rounds 1-3 or with this one; that span is still reported, as
        # config.ms_per_step_own_span: at N = 8 the driver's 20 plies are ~72 us of kernel time per rank and an RCCL barrier
        # is tens of microseconds.)
        "value": total * K / elapsed,
        "unit": "env-steps/s",
        "n_gpus": world,
        "steps": K,
        "warmup": W,
        "ms_per_step": elapsed / K * 1e3,
        "higher_is_better": True,
        "scaling": "weak" if args.boards_per_gpu else "strong",
        "vs_baseline": None,
        "dtype": "int8",
        "data": "synthetic",
        "config": {"workload": f"{total} boards in total = {boards} per GPU x {world} GPU(s) (BASELINE.md C4), masked-random "
                               f"actions, auto-reset, {variant} outputs (state+mask{'' if args.no_obs else '+obs'}"
                               f"+winner+reward+done) every ply, no collective on the step path",
                   "boards_per_gpu": boards, "total_boards": total, "mode": args.mode,
                   "plies_per_launch": args.traj if args.mode == "collect" else 1,
                   "launches_timed": nlaunch * (2 if args.mode == "step2" else 1),
                   "launch": "hipGraph replay" if graphed else "eager launches",
                   # the dominant kernel's mean launch duration on the slowest / fastest rank (HIP events), timed pass
                   "kernel_us_max": max(per_rank_us), "kernel_us_min": min(per_rank_us),
                   # the same K plies played FOUR MORE times behind the contract's pass (rank 0; mean launch duration of each
                   # pass by HIP events): whether the one timed pass was luck shows here
                   "kernel_us_median_of_5": reps[len(reps) // 2] if reps else None,
                   "kernel_us_min_of_5": reps[0] if reps else None, "kernel_us_max_of_5": reps[-1] if reps else None,
                   # each rank's own span, from the synchronize behind the leading barrier to the synchronize behind its last
                   # launch, without the trailing barrier (MAX over ranks): round 4's `ms_per_step`
                   "ms_per_step_own_span": local_elapsed / K * 1e3,
                   # both definitions of the headline under stable names: `value` IS value_contract_span
                   "value_contract_span": total * K / elapsed, "value_own_span": total * K / local_elapsed,
                   # ranks that took part in the barriers / reductions over RCCL (0: none, or a gloo rehearsal)
                   "rccl_ranks": world if (distributed and args.dist_backend == "nccl") else 0,
                   "dist_backend": (args.dist_backend if distributed else None),
                   # the barrier on both sides of the timed span (--span-barrier): the node's ranks meet in shared memory
                   "span_barrier": getattr(args, "span_barrier_used", None),
                   # where the observation / mask trajectory arrays lie (gobblet-rl_amd/placement.py): probe ratio
                   # both / (obs alone + mask alone), ~1.0 = same 96 GiB class of HBM, ~0.8 = different classes;
                   # "unplaced" = some rank's search found no pair in different classes (the headline then runs ~20 % slower)
                   "placement_ratio_min": min(ratios) if ratios else None,
                   "placement_ratio_max": max(ratios) if ratios else None,
                   "placement": (None if not ratios else "unplaced" if max(ratios) >= 0.95 else
                                 "placed" if max(ratios) <= 0.86 else "partly placed")},
        "roofline": roof,
        "detail": {"kernel_us_per_rank": per_rank_us, "trajectory_placement_per_rank": per_rank_placement,
                   "kernel_us_of_5_passes": repeats_us,
                   "sharding": f"contiguous board ranges, {world} shard(s), no collective on the step path",
                   "launch": ("hipGraph replay of the K plies' launches (device-resident ply index; one untimed warm replay of "
                              "the same graph = K more untimed plies before the timed one)") if graphed else "eager"},
    }


def compact_line(full, configs_path):
    """The ONE line of stdout: the contract's keys, flat `roofline` and `cpu_baseline`, nothing nested deeper and no lists."""
    line = {k: v for k, v in full.items() if k not in ("configs", "cpu_baseline", "roofline", "detail", "config", "scale_prediction")}
    line["config"] = dict(full["config"])
    if "configs" in full:
        line["config"]["configs_file"] = configs_path
        line["config"]["configs_recorded"] = len(full["configs"])
    line["roofline"] = {k: full["roofline"].get(k) for k in ROOFLINE_KEYS}
    # where `traffic` comes from (a committed PMC pass keyed by the kernel sources' hash -- not measured in this run), shortened
    src = line["roofline"].get("traffic_source")
    if isinstance(src, str) and len(src) > 140:
        line["roofline"]["traffic_source"] = src[:137] + "..."
    if "cpu_baseline" in full:
        line["cpu_baseline"] = {k: full["cpu_baseline"].get(k) for k in CPU_BASELINE_KEYS}
    if "configs" in full:
        line["configs_compact"] = {"_": "[value, us_per_step, frac of the bound's peak]"}
        for name in COMPACT_CONFIGS:
            rec = full["configs"].get(name)
            if rec:
                line["configs_compact"][name] = [sig(rec.get("value")), sig(rec.get("us_per_step")),
                                                 sig((rec.get("roofline") or {}).get("frac"), 3)]
    for sub in ("config", "roofline", "cpu_baseline"):  # six significant digits are more than any of these numbers carries
        for k, v in line.get(sub, {}).items():
            if isinstance(v, float):
                line[sub][k] = sig(v, 6)
    if "scale_prediction" in full:  # [predicted whole-job env-steps/s, predicted efficiency vs N x this line] per N (see scale_prediction)
        line["scale_prediction"] = {n: [sig(r["value_predicted"]), sig(r["efficiency_predicted"], 3)]
                                    for n, r in full["scale_prediction"].items()}
    text = json.dumps(line)
    if len(text.encode()) > COMPACT_LIMIT:  # never print a line the driver's tail would cut: drop the prose first
        for k in ("timing", "sample", "traffic_source", "accounting"):
            line["roofline"].pop(k, None)
            line.get("cpu_baseline", {}).pop(k, None)
        text = json.dumps(line)
    assert len(text.encode()) <= COMPACT_LIMIT, "bench.py: the contract line outgrew the driver's tail"
    return text


def emit(full, configs_out):
    """Full record (sub-records of the other BASELINE configs, per-rank lists, the CPU baseline's extras) -> a file and stderr;
    the compact contract line -> the LAST line of stdout."""
    path = None
    if configs_out:
        path = configs_out if os.path.isabs(configs_out) else os.path.join(ROOT, configs_out)
        try:
            os.makedirs(os.path.dirname(path), exist_ok=True)
            with open(path, "w") as f:
                json.dump(full, f, indent=1)
        except OSError as e:  # a read-only checkout: stderr still carries the record
            print(f"bench.py: could not write {path}: {e}", file=sys.stderr)
            path = None
    print(json.dumps(full), file=sys.stderr, flush=True)
    print(compact_line(full, os.path.relpath(path, ROOT) if path else None), file=CONTRACT_STDOUT or sys.stdout, flush=True)


CONTRACT_STDOUT = None  # the process's original stdout, once keep_stdout_for_the_line() has pointed fd 1 at stderr


def keep_stdout_for_the_line():
    """Everything this process -- or a library inside it -- writes to fd 1 from now on goes to stderr; only emit()'s contract line
    reaches the real stdout.  RCCL prints a five-line version banner to stdout when it builds its first communicator (seen on
    the pool's image with world 1); a driver that parses stdout must find ONE JSON line there."""
    global CONTRACT_STDOUT
    if CONTRACT_STDOUT is None:
        sys.stdout.flush()
        CONTRACT_STDOUT = os.fdopen(os.dup(1), "w")
        os.dup2(2, 1)


def gpus_of_this_node():
    """GPUs this process tree can use, counted in a CHILD process (torch.cuda.device_count() there: exactly what the ranks will
    see, container device filters and *_VISIBLE_DEVICES included) so that the launcher parent itself imports neither torch nor
    HIP before it starts its children.  None when the count cannot be had (the ranks' own check then fires)."""
    import subprocess
    try:
        out = subprocess.run([sys.executable, "-c", "import torch; print(torch.cuda.device_count())"], capture_output=True,
                             text=True, timeout=300)
        return int(out.stdout.strip().splitlines()[-1])
    except Exception:  # noqa: BLE001
        return None


def spawn_ranks(n):
    """`python bench.py --gpus N` without a launcher: start the N rank processes ourselves -- torch.distributed.run as a
    CHILD process; this parent imports neither torch nor HIP (the devices are counted by a child too) -- and pass its exit code on.
    Rank 0's JSON line is the child's stdout, i.e. ours."""
    import socket
    import subprocess
    if "--share-device" not in sys.argv:
        ndev = gpus_of_this_node()
        if ndev is not None and n > ndev:
            print(f"bench.py: --gpus {n} but this node has {ndev} GPU(s): one rank per GPU is the contract "
                  f"(--share-device + --dist-backend gloo rehearses more ranks on one card)", file=sys.stderr)
            return 2
    with socket.socket() as sock:  # a free port for the rendezvous
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # the pool's driver supports dmabuf IPC only (RCCL needs it)
    env.setdefault("OMP_NUM_THREADS", "1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args.gpus))
    import torch

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        print(f"bench.py: --gpus {args.gpus} but the launcher started {world} rank(s): measuring {world}", file=sys.stderr)
        args.gpus = world
    # FAIL FAST, before any rendezvous: more ranks than devices (e.g. `--gpus 8` under a launcher on a one-GPU box) would
    # otherwise hang in init_process_group / die in set_device on some ranks while the others wait for them
    ndev = torch.cuda.device_count()  # (does not initialise the GPU)
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", world))  # (a multi-node launch: this node's ranks only)
    if not args.share_device and local_world > max(ndev, 0):
        sys.exit(f"bench.py: {local_world} rank(s) but this node has {ndev} GPU(s): one rank per GPU is the contract "
                 f"(--share-device + --dist-backend gloo rehearses more ranks on one card)")
    if not torch.cuda.is_available():
        sys.exit("bench.py needs an MI355X (there is no CPU fallback for the product path)")
    if args.share_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1 or os.environ.get("GBL_BENCH_FORCE_DIST"):  # (the env knob rehearses the RCCL path at world 1)
        import torch.distributed as dist
        keep_stdout_for_the_line()
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)  # RCCL; used only for barriers + small reductions of timings
        else:
            dist.init_process_group("gloo")

    import gobblet_rl_amd as G

    if args.boards_per_gpu:
        boards, env_base, total = args.boards_per_gpu, rank * args.boards_per_gpu, args.boards_per_gpu * world
    else:
        env_base, boards = G.shard_bounds(args.boards, world, rank)
        total = args.boards
    K, W = args.steps, args.warmup
    args.traj = auto_traj(boards, K, args.traj, no_obs=args.no_obs)
    p = Pipeline(G, torch, boards, env_base, dev, no_obs=args.no_obs, mode=args.mode, traj=args.traj,
                 placement=args.placement)
    nlaunch = len(p.plan(K))
    p.eager(W)  # warm-up plies (untimed): decorrelate game phases, warm caches / code objects
    torch.cuda.synchronize(dev)
    if dist is not None:
        # the communicator's one-off costs (RCCL builds its rings inside the first collective) stay out of the timed span: the
        # barrier that opens it is then an ordinary one
        dist.barrier()
        dist.barrier()

    # A timed run of ONE launch (the driver's 20 plies) is launched eagerly: a graph of one kernel node + the counter node starts
    # no sooner and carries the counter node inside the timed span (131 072 boards x 20 plies: 90 vs 99 us of wall clock,
    # scripts/experiments/launch_overhead.py); several launches are replayed as one hipGraph.  Either way the K plies are played ONCE untimed
    # through the very same code path first (the first launch of an instantiated graph carries one-off costs, and so does the first
    # pass of the host code: 30 us on a 560 us region), then timed.
    use_graph = bool(args.graph) and nlaunch > 1
    graph = p.capture(K) if use_graph else None
    ev = p.events(1 if use_graph else nlaunch)
    stream = p.nat.current_stream(dev)

    # eager gbl_collect launches take the ply index by value (what BatchedGobblet.collect does without device_ply()): the launch
    # that would move a device-resident index on is not needed, and not in the span (scripts/experiments/span_overhead.py: 7 of its 22 us)
    by_value = graph is None and args.mode == "collect"

    def play_k():
        if graph is not None:
            ev[0][0].record()
            graph.replay()        # (its last node moves the device-resident ply index on)
            ev[0][1].record()
        else:
            for i, (off, plies) in enumerate(p.plan(K)):
                p.enqueue(off, plies, stream, ev[i], by_value=by_value)

    def bookkeeping():
        if by_value:
            p.played(K)           # (host arithmetic)
        elif graph is None:       # eager, one ply per launch: the launch that moves the device-resident ply index on
            p.advance(K, stream)

    span_barrier, lb = None, None
    args.span_barrier_used = None
    if dist is not None:
        local = int(os.environ.get("LOCAL_WORLD_SIZE", world)) == world
        if args.span_barrier == "local" and local:
            lb = LocalBarrier.create(dist, rank, world)
        if lb is not None:
            span_barrier, args.span_barrier_used = lb.wait, "node-local shared memory"
        else:
            span_barrier, args.span_barrier_used = dist.barrier, "torch.distributed (%s)" % args.dist_backend

    def pass_kernel_us():         # mean launch duration of the dominant kernel over the pass just played (HIP events)
        if graph is None:
            return sum(a.elapsed_time(b) for a, b in ev) / nlaunch * 1e3
        return ev[0][0].elapsed_time(ev[0][1]) / nlaunch * 1e3

    play_k()                      # untimed rehearsal: K more warm plies
    bookkeeping()
    torch.cuda.synchronize(dev)
    if span_barrier is not None:
        span_barrier()            # (rehearsed too)
        span_barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    play_k()
    bookkeeping()                 # (inside the timed region, as in the graph)
    torch.cuda.synchronize(dev)
    local_elapsed = time.perf_counter() - t0  # this rank's own span (reported beside the contract's)
    if span_barrier is not None:
        span_barrier()
    torch.cuda.synchronize(dev)
    elapsed = time.perf_counter() - t0        # the contract's span: barrier + synchronize on both sides

    # dominant-kernel time for the roofline: HIP events on the launch stream
    plies_timed = K
    if graph is None:
        kernel_s = sum(a.elapsed_time(b) for a, b in ev) / 1e3
        launches, timing = nlaunch, "HIP event pair around every launch of the dominant kernel"
    elif args.mode != "step2":
        kernel_s = ev[0][0].elapsed_time(ev[0][1]) / 1e3
        launches, timing = nlaunch, "HIP events around the graph replay of all launches (includes kernel boundaries)"
    else:
        # two kernels per ply in the graph: time gbl_step on its own with event pairs, eagerly, after the timed region
        plies_timed = launches = min(K, 64)
        ev2 = p.events(launches)
        p.eager(launches, ev2)
        torch.cuda.synchronize(dev)
        kernel_s = sum(a.elapsed_time(b) for a, b in ev2) / 1e3
        timing = "HIP event pair around each of %d eager gbl_step launches after the timed replay" % launches
    mean_kernel_s = kernel_s / launches
    # the headline's timed sample may be ONE launch: play the same K plies four more times (rank 0 reports the spread)
    repeats_us = []
    if args.mode != "step2":
        repeats_us.append(pass_kernel_us())
        for _ in range(4):
            play_k()
            bookkeeping()
            torch.cuda.synchronize(dev)
            repeats_us.append(pass_kernel_us())
    p.settle(stream)  # (plies launched with the index by value: the device-resident index catches up)
    per_rank_us = [mean_kernel_s * 1e6]
    mine_pl = p.traj["_placement"] if p.traj is not None else None
    per_rank_placement = [mine_pl]
    if dist is not None:
        cpu = args.dist_backend != "nccl"
        tt = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if cpu else dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
        # per rank: kernel time, and where its trajectory arrays ended up (probe ratio, probes, GiB held while searching, its
        # cap, spread flag; -1 = no search) -- one small all-gather of numbers, after the timed region
        pl = mine_pl or {}
        mine = torch.tensor([mean_kernel_s * 1e6, pl.get("ratio", -1.0), len(pl.get("probes", ())), pl.get("held_gib", -1.0),
                             pl.get("cap_gib", -1.0), 1.0 if pl.get("spread") else 0.0, local_elapsed * 1e6],
                            dtype=torch.float64, device="cpu" if cpu else dev)
        allk = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allk, mine)
        rows = [[float(v) for v in x.tolist()] for x in allk]
        per_rank_us = [r[0] for r in rows]
        local_elapsed = max(r[6] for r in rows) / 1e6
        per_rank_placement = [{"ratio": r[1], "probes": int(r[2]), "held_gib": r[3], "cap_gib": r[4], "spread": bool(r[5])}
                              if r[1] >= 0 else None for r in rows]
        per_rank_placement[0] = mine_pl if rank == 0 else per_rank_placement[0]

    if rank == 0:
        roof = p.kernel_roofline(kernel_s, plies_timed, launches, timing)
        # (the counters were taken on launches of exactly T plies; the timed launches may end with a shorter one)
        attach_traffic(roof, p, min(args.traj, K) if args.mode == "collect" else 1)
        full = contract_record(args, p, roof, total, boards, world, K, W, elapsed, local_elapsed, nlaunch, graph is not None,
                               per_rank_us, per_rank_placement, dist is not None, repeats_us)
        if world == 1 and not args.no_configs:
            cfg = {}
            for name, (n, k, noobs, mode) in CONFIG_RECORDS.items():
                cfg[name] = short_run(G, torch, dev, n, k, max(W, 64) if name == "headline_recipe_1048576" else W, no_obs=noobs,
                                      mode=mode, traj=32 if name.endswith("_T32") else auto_traj(n, k, no_obs=noobs),
                                      placement=args.placement)
            # an external policy's ply + the masked-random reply in one launch, at the C3 / C4-shard sizes (against single_ply_*)
            for n in (131072, 262144):
                cfg[f"step_reply_{n}"] = step_reply_run(G, torch, dev, n, 200, W)
            cfg["c5_greedy_65536"] = greedy_run(G, torch, dev)
            cfg["greedy_collect_65536"] = greedy_collect_run(G, torch, dev)
            full["configs"] = cfg
            if args.mode == "collect" and not args.boards_per_gpu and total % 8 == 0:
                full["scale_prediction"] = scale_prediction(G, torch, dev, total, K, W, full["value"], args.placement)
        if world == 1 and not args.no_cpu_baseline:
            full["cpu_baseline"] = cpu_baseline(boards, W, args.cpu_seconds)
            ref = full["cpu_baseline"].get("greedy_depth2")
            if ref and "configs" in full:
                # the reference's work per decision (counted by the CPU restatement on the same
                # stationary mix) x the measured decisions/s
                c5 = full["configs"]["c5_greedy_65536"]
                c5["reference_work_per_decision"] = {k: ref[k] for k in ("legality_tests_per_decision", "leaf_evaluations_per_decision")}
                c5["legality_tests_per_s_equivalent"] = c5["value"] * ref["legality_tests_per_decision"]
                c5["leaf_evaluations_per_s_equivalent"] = c5["value"] * ref["leaf_evaluations_per_decision"]
                c5["cpu_port_decisions_per_s_1core"] = ref["decisions_per_s_1core"]
        emit(full, args.configs_out)
    if lb is not None:
        lb.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
