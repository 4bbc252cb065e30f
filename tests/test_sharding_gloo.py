"""world_size-2 and -8 `gloo` tests of the N>1 path on CPU: contiguous shards, sampler keyed by the global
board id (results independent of the number of shards), no collective on the step path, tallies
summed afterwards.  The per-shard compute is the device code compiled for the host (tests/emu),
standing in for the GPU each rank would own."""
import os
import tempfile

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import gobblet_rl_amd as G
import oracle
from tests import emu

TOTAL, PLIES, SEED = 1000 + 37, 30, 11


def _worker(rank, world, initfile, outdir):
    dist.init_process_group("gloo", init_method=f"file://{initfile}", rank=rank, world_size=world)
    start, count = G.shard_bounds(TOTAL, world, rank)
    s, tm, dn = oracle.batch_reset(count)
    out = emu.rollout(s, tm, dn, SEED, start, 0, PLIES)          # env_base = first global board of the shard
    tallies = G.reduce_counters(torch.from_numpy(out["counters"]))  # the only communication, after the run
    np.savez(os.path.join(outdir, f"r{rank}.npz"), start=start, count=count, state=s, to_move=tm,
             mask=out["mask"], tallies=tallies.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_shard_bounds():
    for total, world in [(1 << 20, 8), (1037, 2), (5, 8), (64, 1)]:
        spans = [G.shard_bounds(total, world, r) for r in range(world)]
        assert spans[0][0] == 0 and sum(c for _, c in spans) == total
        assert all(spans[i][0] + spans[i][1] == spans[i + 1][0] for i in range(world - 1))
        assert max(c for _, c in spans) - min(c for _, c in spans) <= 1
    with pytest.raises(ValueError):
        G.shard_bounds(10, 2, 2)


@pytest.mark.parametrize("world", [2, 8])
def test_gloo_ranks_equal_single_shard(world):
    """Two ranks, and EIGHT -- the shape of BASELINE config 4 (one rank per GPU of a node; here eight CPU processes over gloo):
    ragged shards (1037 boards over 8 ranks: 130 and 129), the rendezvous, the one all-reduce after the run."""
    with tempfile.TemporaryDirectory() as d:
        initfile = os.path.join(d, "init")
        mp.spawn(_worker, args=(world, initfile, d), nprocs=world, join=True)
        parts = [np.load(os.path.join(d, f"r{r}.npz")) for r in range(world)]
    s, tm, dn = oracle.batch_reset(TOTAL)
    ref = oracle.batch_rollout(s, tm, dn, SEED, 0, 0, PLIES, threads=4)
    assert int(parts[0]["start"]) == 0
    assert all(int(parts[r + 1]["start"]) == int(parts[r]["start"]) + int(parts[r]["count"]) for r in range(world - 1))
    assert np.array_equal(np.concatenate([p["state"] for p in parts]), s)
    assert np.array_equal(np.concatenate([p["to_move"] for p in parts]), tm)
    assert np.array_equal(np.concatenate([p["mask"] for p in parts]), ref["mask"])
    for p in parts:  # every rank holds the global tallies
        assert np.array_equal(p["tallies"], ref["counters"])


def _gpu_worker(rank, world, initfile, outdir):
    dist.init_process_group("gloo", init_method=f"file://{initfile}", rank=rank, world_size=world)
    env = G.make_shard(TOTAL, rank, world, "cuda:0", auto_reset=True, seed=SEED)  # both ranks share the one GPU here
    env.rollout(PLIES, count=True)
    tallies = G.reduce_counters(env.counters.cpu())
    np.savez(os.path.join(outdir, f"g{rank}.npz"), state=env.squares.cpu().numpy(), mask=env.action_mask.cpu().numpy(),
             obs=env.observation.cpu().numpy(), tallies=tallies.numpy(), base=env.env_base)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.gpu
def test_two_rank_gloo_on_gpu_equals_single_shard():
    """The same with the real kernels: two ranks (sharing the one GPU of the test box), contiguous shards,
    no collective on the step path, tallies summed afterwards == one shard holding every board."""
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_gpu_worker, args=(2, os.path.join(d, "init"), d), nprocs=2, join=True)
        parts = [np.load(os.path.join(d, f"g{r}.npz")) for r in range(2)]
    s, tm, dn = oracle.batch_reset(TOTAL)
    ref = oracle.batch_rollout(s, tm, dn, SEED, 0, 0, PLIES, threads=4)
    assert int(parts[1]["base"]) == len(parts[0]["state"])
    assert np.array_equal(np.concatenate([p["state"] for p in parts]), s)
    assert np.array_equal(np.concatenate([p["mask"] for p in parts]), ref["mask"])
    assert np.array_equal(np.concatenate([p["obs"] for p in parts]), ref["obs"])
    for p in parts:
        assert np.array_equal(p["tallies"], ref["counters"])


def _barrier_worker(rank, world, initfile, outdir):
    import importlib.util
    import time
    spec = importlib.util.spec_from_file_location("bench", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    dist.init_process_group("gloo", init_method=f"file://{initfile}", rank=rank, world_size=world)
    lb = bench.LocalBarrier.create(dist, rank, world)
    assert lb is not None
    before, after = [], []
    for i in range(50):
        if (i + rank) % world == 0:
            time.sleep(0.002)            # a different straggler every round
        before.append(time.monotonic())  # (CLOCK_MONOTONIC: one clock for all processes of the host)
        lb.wait()
        after.append(time.monotonic())
    np.savez(os.path.join(outdir, f"b{rank}.npz"), before=np.array(before), after=np.array(after))
    lb.close()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 8])
def test_span_barrier_of_the_node(world):
    """bench.py's barrier on both sides of the timed span (LocalBarrier: generation counters of the node's ranks in shared
    memory): nobody leaves round i before everybody has arrived at it, fifty rounds with a different straggler each."""
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_barrier_worker, args=(world, os.path.join(d, "init"), d), nprocs=world, join=True)
        parts = [np.load(os.path.join(d, f"b{r}.npz")) for r in range(world)]
    before = np.stack([p["before"] for p in parts])
    after = np.stack([p["after"] for p in parts])
    assert (after.min(axis=0) >= before.max(axis=0)).all()
