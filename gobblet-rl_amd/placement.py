"""Where the two large trajectory arrays of ``gbl_collect`` lie in HBM.

Measured on MI355X (scripts/placement_map.py, placement_map2.py, placement_probe_check.py): the 288 GiB
of device memory fall into three classes of 96 GiB -- by the size, the three die groups of the 12-high HBM3E stacks --
and two write streams inside ONE class do not overlap: ``gbl_collect`` then takes the sum of what its observation
stream and its mask stream take alone (33-34 us per ply at 2^20 boards), against 27 us when the two arrays lie in
different classes (6.85 TB/s of trajectory writes, the rate of a plain ``fill_``).  Consecutive allocations of a fresh
process usually come from the same class, which made the trajectory stream 20 % slower in three runs out of four.

A process cannot see physical addresses, so the class is measured: ``gbl_placement_probe`` replays the kernel's store
pattern on two buffers (both, a alone, b alone; ratio both / (a + b) ~1.0 inside one class, ~0.80 across two, in between
when an array itself straddles two classes).  ``spread_pair`` keeps a small pool of candidates for either array --
every new candidate allocated behind an 8 GiB spacer, so that it comes from another of the driver's physical blocks
(the class changes every 8-64 GiB along a process's allocations) -- probes the new candidate against the pool of the
other array, stops at the first clean pair, and releases everything else (spacers included) at the end."""
from __future__ import annotations

import ctypes as C
import time

import torch

from . import _native as nat

GIB = 1 << 30
MIN_BYTES = 64 << 20       # below this a probe says nothing (and the arrays live in the 256 MiB Infinity Cache anyway)
ACCEPT_RATIO = 0.83        # stop searching at a pair this good (us_both / (us_a + us_b)); 0.86 already costs 3 %
SPREAD_RATIO = 0.93        # reported as "spread" below this
SAME_RATIO = 0.96          # above this the pair simply shares a class
STEP_BYTES = 8 * GIB
MAX_SKIP_BYTES = 96 * GIB  # spacers held at most, transiently (a class is 96 GiB)
MAX_PROBES = 16
RESERVE_BYTES = 4 * GIB    # never take the device's last few GiB for spacers


def probe(a: torch.Tensor, b: torch.Tensor, slot_boards: int = 0, plies: int = 0) -> tuple[float, float, float]:
    """(us_both, us_a, us_b) of ``gbl_placement_probe`` on two device tensors; slot_boards / plies: the geometry of the
    time-major trajectory they will hold (0: four slots over the smaller one).  OVERWRITES both with zeros."""
    both, ua, ub = C.c_float(), C.c_float(), C.c_float()
    with torch.cuda.device(a.device):  # (the probe creates events and launches: on the arrays' device)
        nat.check(nat.lib().gbl_placement_probe(a.data_ptr(), a.numel() * a.element_size(), b.data_ptr(),
                                                b.numel() * b.element_size(), int(slot_boards), int(plies),
                                                C.byref(both), C.byref(ua), C.byref(ub),
                                                nat.current_stream(a.device)), "gbl_placement_probe")
    return both.value, ua.value, ub.value


def spread_pair(make_a, make_b, step_bytes: int = STEP_BYTES, max_skip_bytes: int = MAX_SKIP_BYTES,
                max_probes: int = MAX_PROBES, slot_boards: int = 0, plies: int = 0):
    """``(a, b) = (make_a(), make_b())`` placed so that writes to ``a`` and to ``b`` overlap.  Returns (a, b, info);
    info records every probe.  Both arrays come back zero-filled.  If no pair is clean, the best one seen is returned."""
    t0 = time.perf_counter()
    pool = {"a": [make_a()], "b": [make_b()]}
    dev = pool["a"][0].device
    tried, spacers, skipped = [], [], 0
    best = (None, 0, 0)  # ratio, index into pool a, index into pool b

    def try_pair(ia, ib):
        nonlocal best
        us_both, us_a, us_b = probe(pool["a"][ia], pool["b"][ib], slot_boards, plies)
        ratio = us_both / max(us_a + us_b, 1e-9)
        tried.append(round(ratio, 3))
        if best[0] is None or ratio < best[0]:
            best = (ratio, ia, ib)

    try_pair(0, 0)
    grow = "b"  # candidates are added alternately: a new mask array first (the smaller one)
    while best[0] > ACCEPT_RATIO and len(tried) < max_probes:
        nbytes = sum(t.numel() * t.element_size() for t in (pool["a"][0], pool["b"][0]))
        free, _ = torch.cuda.mem_get_info(dev)
        if skipped + step_bytes > max_skip_bytes or free < step_bytes + RESERVE_BYTES + nbytes:
            break
        try:
            spacers.append(torch.empty(step_bytes, dtype=torch.uint8, device=dev))
            pool[grow].append(make_a() if grow == "a" else make_b())
        except torch.OutOfMemoryError:
            break
        skipped += step_bytes
        new, other = len(pool[grow]) - 1, "b" if grow == "a" else "a"
        # against the other array's first candidate; against the rest only if that pair is neither clean nor a plain
        # conflict (an array that straddles two classes), since all members of a pool probed alike so far
        for k in range(len(pool[other])):
            if best[0] <= ACCEPT_RATIO or len(tried) >= max_probes or (k > 0 and tried[-1] > SAME_RATIO):
                break
            try_pair(*((new, k) if grow == "a" else (k, new)))
        grow = other
    ratio, ia, ib = best
    a, b = pool["a"][ia], pool["b"][ib]
    a.zero_(); b.zero_()  # (a probe writes only zeros, but say so explicitly)
    released = len(spacers) + len(pool["a"]) + len(pool["b"]) - 2
    pool.clear(); spacers.clear()
    if released:
        torch.cuda.empty_cache()  # hand the spacers and rejected candidates back to the driver
    return a, b, {"spread": bool(ratio <= SPREAD_RATIO), "ratio": round(ratio, 3), "probes": tried,
                  "skipped_gib": skipped // GIB, "seconds": round(time.perf_counter() - t0, 3)}
