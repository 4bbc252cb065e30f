#!/usr/bin/env python3
"""Does the trajectory stream's rate depend on HOW its buffers were allocated?  The same gbl_collect graph on
trajectory arrays from torch's caching allocator, from plain hipMalloc and from hipExtMallocWithFlags (fine-grained,
uncached), all in one process."""
import ctypes as C
import os
import re
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gobblet_rl_amd as G  # noqa: E402

if os.environ.get("GOBBLET_HIP_LIB"):  # an experiment's own build of the library (scripts/build_variant.sh)
    G._native.use_library(os.environ["GOBBLET_HIP_LIB"])

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
T = int(sys.argv[2]) if len(sys.argv) > 2 else 8
nat, L = G._native, G._native.lib()
torch.zeros(1, device="cuda:0")
hip_path = [ln.split()[-1] for ln in open("/proc/self/maps") if re.search(r"libamdhip64\.so", ln)][0]
hip = C.CDLL(hip_path)
hip.hipExtMallocWithFlags.argtypes = [C.POINTER(C.c_void_p), C.c_size_t, C.c_uint]
hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
hip.hipFree.argtypes = [C.c_void_p]
hip.hipMemset.argtypes = [C.c_void_p, C.c_int, C.c_size_t]
env = G.BatchedGobblet(n, "cuda:0", auto_reset=True, seed=0)
env.rollout(64)
slot = -(-n // 128) * 128
sizes = {"actions": 4, "winner": 1, "rewards": 2, "done": 1, "to_move": 1, "action_mask": 54, "observation": 117}
ctr = torch.zeros(1, dtype=torch.int32, device="cuda:0")
launches = max(2, 256 // T)


def alloc(flavour):
    ptrs, keep = {}, []
    for k, b in sizes.items():
        nbytes = T * slot * b
        if flavour == "torch":
            t = torch.zeros(nbytes, dtype=torch.uint8, device="cuda:0")
            keep.append(t)
            ptrs[k] = t.data_ptr()
        else:
            p = C.c_void_p()
            rc = hip.hipMalloc(C.byref(p), nbytes) if flavour == "hipMalloc" else hip.hipExtMallocWithFlags(
                C.byref(p), nbytes, {"finegrained": 1, "uncached": 3}[flavour])
            assert rc == 0, (flavour, rc)
            hip.hipMemset(p, 0, nbytes)
            keep.append(p)
            ptrs[k] = p.value
    return ptrs, keep


def measure(P):
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        s = nat.current_stream(torch.device("cuda:0"))
        for i in range(launches):
            nat.check(L.gbl_collect(env.squares.data_ptr(), env.to_move.data_ptr(), env.done.data_ptr(), P["actions"], P["winner"],
                                    P["rewards"], P["done"], P["to_move"], P["action_mask"], P["observation"], n, slot, 64, 0, 0,
                                    i * T, ctr.data_ptr(), T, 0, None, None, s))
        nat.check(L.gbl_counter_add(ctr.data_ptr(), launches * T, s))
    g.replay()
    torch.cuda.synchronize()
    us = []
    for _ in range(5):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); g.replay(); b.record()
        torch.cuda.synchronize()
        us.append(a.elapsed_time(b) * 1e3 / (launches * T))
    return statistics.median(us)


held = []
hold = len(sys.argv) > 3 and sys.argv[3] == "hold"
for rnd in range(3):
    for flavour in ("torch", "hipMalloc", "finegrained", "hipMalloc", "torch"):
        P, keep = alloc(flavour)
        us = measure(P)
        print(f"round {rnd} {flavour:12s}: {us:7.2f} us/ply   obs @ {P['observation']:#014x}  mask @ {P['action_mask']:#014x}  "
              f"actions @ {P['actions']:#014x}", flush=True)
        if hold:
            held.append(keep)
        else:
            for k in keep:
                if not torch.is_tensor(k):
                    hip.hipFree(k)
            del keep, P
            torch.cuda.empty_cache()
