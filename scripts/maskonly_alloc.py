#!/usr/bin/env python3
"""MASK_ONLY gbl_collect at 2^20 boards: does it matter where the ONE large array (the mask trajectory) comes from?
torch's caching allocator against the head of a hipMalloc block of its own (placement.DeviceBlock), T = 8 and 32 plies per launch;
hipGraph of 256 plies, HIP events.    python scripts/maskonly_alloc.py [boards]"""
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gobblet_rl_amd as G  # noqa: E402
from gobblet_rl_amd import placement  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
nat, L = G._native, G._native.lib()
dev = torch.device("cuda:0")
env = G.BatchedGobblet(n, dev, auto_reset=True, seed=0, with_observation=False)
env.rollout(64)
ctr = torch.zeros(1, dtype=torch.int32, device=dev)
for T in (8, 32):
    buf = env.trajectory_buffers(T, placement="any")
    f = buf["_full"]
    cells = f["action_mask"].numel()
    arrays = {"torch allocator": f["action_mask"],
              "own hipMalloc block": placement.DeviceBlock(placement.block_bytes(cells), dev).tensor()[:cells].view(torch.int8).view(f["action_mask"].shape),
              "own block + 1 MiB": placement.DeviceBlock(placement.block_bytes(cells) + (2 << 20), dev).tensor()[1 << 20:(1 << 20) + cells].view(torch.int8).view(f["action_mask"].shape)}
    for name, arr in arrays.items():
        launches = 256 // T
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            s = nat.current_stream(dev)
            for i in range(launches):
                nat.check(L.gbl_collect(env.squares.data_ptr(), env.to_move.data_ptr(), env.done.data_ptr(), f["actions"].data_ptr(),
                                        f["winner"].data_ptr(), f["rewards"].data_ptr(), f["done"].data_ptr(), f["to_move"].data_ptr(),
                                        arr.data_ptr(), None, n, buf["_ply_stride"], buf["_tile_stride"], 0, 0, i * T, ctr.data_ptr(), T, 0,
                                        None, None, s))
            nat.check(L.gbl_counter_add(ctr.data_ptr(), launches * T, s))
        g.replay()
        torch.cuda.synchronize()
        r = []
        for _ in range(7):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); g.replay(); b.record()
            torch.cuda.synchronize()
            r.append(a.elapsed_time(b) * 1e3 / 256)
        print(f"boards {n} MASK_ONLY T={T:2d} mask array from {name:22s} (address {arr.data_ptr():#x}): median {statistics.median(r):6.2f} us/ply  min {min(r):6.2f}"
              f"  = {n * (61 + 57 / T) / statistics.median(r) / 8e6:.3f} of the HBM peak", flush=True)
