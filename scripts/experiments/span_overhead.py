#!/usr/bin/env python3
"""What the contract's timed span of ONE eager K-ply launch consists of besides the kernel (the driver's `--steps 20`; at 8 GPUs
a shard is 131 072 boards and the kernel lasts ~73 us): the host's launch path, the launch that advances the device-resident
ply index, and how the host waits for the end.  Variants, wall clock around each (median of REPS), kernel time by HIP events:
    sync            events + gbl_collect + gbl_counter_add + torch.cuda.synchronize          (bench.py until round 5)
    no-advance      the ply index passed by value: no second launch
    poll            as sync, but the host polls the stop event before it calls synchronize
    no-advance+poll both
    prebound args   no-advance with the 22 ctypes arguments converted once (nothing: the path is the two event records and the launch)
  SPAN_SPIN=1|2|4 sets hipDeviceScheduleSpin / Yield / BlockingSync first (nothing either: the wait already spins)
  python scripts/experiments/span_overhead.py [BOARDS] [PLIES]"""
import os
import statistics
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import importlib.util  # noqa: E402

spec = importlib.util.spec_from_file_location("bench", os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "bench.py"))
B = importlib.util.module_from_spec(spec)
spec.loader.exec_module(B)
import gobblet_rl_amd as G  # noqa: E402

if os.environ.get("SPAN_SPIN"):  # hipDeviceScheduleSpin (1) / Yield (2) / BlockingSync (4): how hipDeviceSynchronize waits
    import ctypes
    hip = ctypes.CDLL("libamdhip64.so")
    torch.cuda.init()
    print("hipSetDeviceFlags ->", hip.hipSetDeviceFlags(int(os.environ["SPAN_SPIN"])), flush=True)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 131072
K = int(sys.argv[2]) if len(sys.argv) > 2 else 20
REPS = 25
dev = torch.device("cuda:0")
p = B.Pipeline(G, torch, n, 0, dev, traj=K)
p.eager(64)
torch.cuda.synchronize()
stream = p.nat.current_stream(dev)
ev = p.events(1)[0]
T, P, env = p.TP, p.P, p.env
ply = [64]


def launch(by_value):
    ev[0].record()
    rc = p.lib.gbl_collect(P["sq"], P["tm"], P["dn"], T["ac"], T["wi"], T["rw"], T["dn"], T["tm"], T["mk"], T["ob"], n,
                           p.traj["_ply_stride"], p.traj["_tile_stride"], env.seed, env.env_base, ply[0] if by_value else 0,
                           None if by_value else p.ctr.data_ptr(), K, 0, None, None, stream)
    assert rc == 0
    ev[1].record()


import ctypes as C  # noqa: E402

_sig = G._native.SIGNATURES["gbl_collect"][1]
_vals = [P["sq"], P["tm"], P["dn"], T["ac"], T["wi"], T["rw"], T["dn"], T["tm"], T["mk"], T["ob"], n, p.traj["_ply_stride"],
         p.traj["_tile_stride"], env.seed, env.env_base, 0, None, K, 0, None, None, stream]
_cargs = [t(v) if v is not None else None for t, v in zip(_sig, _vals)]   # converted once
_rec0, _rec1, _fn = ev[0].record, ev[1].record, p.lib.gbl_collect


def launch_prebound():
    _cargs[15] = ply[0]
    _rec0()
    rc = _fn(*_cargs)
    _rec1()
    assert rc == 0


def run(by_value, poll):
    t0 = time.perf_counter()
    if by_value == "prebound":
        launch_prebound()
    else:
        launch(by_value)
    t1 = time.perf_counter()
    if not by_value:
        p.advance(K, stream)
    t2 = time.perf_counter()
    if poll:
        while not ev[1].query():
            pass
    torch.cuda.synchronize(dev)
    t3 = time.perf_counter()
    ply[0] += K
    return (t1 - t0) * 1e6, (t2 - t1) * 1e6, (t3 - t2) * 1e6, (t3 - t0) * 1e6, ev[0].elapsed_time(ev[1]) * 1e3


for name, by_value, poll in (("sync", False, False), ("no-advance", True, False), ("poll", False, True), ("no-advance+poll", True, True),
                             ("prebound args", "prebound", False), ("sync", False, False)):
    for _ in range(3):
        run(by_value, poll)
    rows = [run(by_value, poll) for _ in range(REPS)]
    med = [statistics.median(r[i] for r in rows) for i in range(5)]
    print(f"boards {n} x {K} plies  {name:16s}: launch path {med[0]:6.1f} us  advance {med[1]:5.1f}  wait {med[2]:6.1f}  "
          f"span {med[3]:6.1f}  kernel {med[4]:6.1f}  span - kernel {med[3] - med[4]:5.1f}", flush=True)
