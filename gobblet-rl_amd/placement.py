"""Where the two large trajectory arrays of ``gbl_collect`` lie in HBM.

Measured on MI355X (scripts/experiments/placement_map.py, placement_map2.py, placement_probe_check.py): the 288 GiB
of device memory fall into three classes of 96 GiB -- by the size, the three die groups of the 12-high HBM3E stacks --
and two write streams inside ONE class do not overlap: ``gbl_collect`` then takes the sum of what its observation
stream and its mask stream take alone (33-34 us per ply at 2^20 boards), against 27 us when the two arrays lie in
different classes (6.85 TB/s of trajectory writes, the rate of a plain ``fill_``).  Consecutive allocations of a fresh
process usually come from the same class, which made the trajectory stream 20 % slower in three runs out of four.

A process cannot see physical addresses, so the class is measured: ``gbl_placement_probe`` replays the kernel's store
pattern on two buffers (both, a alone, b alone; ratio both / (a + b) ~1.0 inside one class, ~0.80 across two, in between
when an array itself straddles two classes).  ``spread_pair`` carves either array from the start of a BLOCK of its own
(at least 2 GiB, one ``hipMalloc`` through the library's ``gbl_block_alloc`` -- outside torch's caching allocator, so
that a rejected block goes straight back to the driver when it is dropped and nobody's cache is flushed): small
allocations of a process all come from one neighbourhood of physical memory whatever is allocated in between (twelve
candidates behind 8 GiB spacers each: the same ratio twelve times), whereas blocks of 2 GiB and more come from all over
the device and change class every few blocks.  It keeps a small pool of blocks for either array, probes a new one
against the other array's first -- leaving a gap of 2, 4, 8 ... GiB in front of it after every plain conflict, because
on a fresh device consecutive blocks can stay inside one 96 GiB class -- stops at the first clean pair and frees the
other blocks and the gaps at the end.

What a caller that shares the device can rely on (round 3):
  * the pair as torch's allocator places it is probed first and stays in the race: if it is clean it is used as it is (no
    block, nothing pinned), and a block pair replaces it only if it is clearly better -- the result is never worse than
    the caller's own placement (with 200 GiB of the device taken the capped search found nothing in three runs out of
    three while the allocator's pair was clean, profiles/r03/placement_ab.txt);
  * the search never holds more than ``MAX_HOLD_BYTES`` (64 GiB) nor more than a quarter of the memory that was free
    when it started (the two arrays' own blocks always count), and leaves ``RESERVE_BYTES`` untouched -- EXCEPT with
    ``far=True`` (opt-in: a caller that owns the device, e.g. bench.py), whose candidates lie behind a transient gap of
    64 / 96 / 128 GiB that is held for the duration of two ``hipMalloc`` calls (never while probing) and recorded as
    ``transient_peak_gib``; the default (``far=None``) tries them only on a device with ``FAR_MIN_FREE_BYTES`` (160 GiB)
    free, ``far=False`` never (``collect()``'s implicit staging buffers);
  * an allocation the device refuses ENDS the search (the best pair seen so far is used); if not even the two arrays'
    own blocks fit the plain pair is kept (without one, ``PlacementUnavailable`` is raised) -- the reason is recorded;
  * ``torch.cuda.empty_cache()`` is never called: rejected blocks are ``hipFree``d, torch's cache is left alone -- which also
    means that when a block pair replaces the allocator's own pair, that pair's memory stays in torch's cache (``torch_cache_gib``
    in the record: 1.4 GiB at 2^20 boards x 8 plies) on top of the winner's blocks;
  * a block whose last owner goes away during a stream capture is parked and freed later (``hipFree`` synchronises).
"""
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
MIN_BLOCK_BYTES = 2 * GIB  # an array is carved from a block of its own of at least this size
BLOCK_GRANULE = 2 << 20    # blocks are whole 2 MiB pages
MAX_HOLD_BYTES = 64 * GIB  # blocks and gaps held at most while searching (a class is 96 GiB) ...
FREE_FRACTION = 4          # ... and never more than 1 / FREE_FRACTION of the memory free at entry
MAX_SKIP_BYTES = 32 * GIB  # the largest single gap
MAX_PROBES = 16
RESERVE_BYTES = 4 * GIB    # never take the device's last few GiB for the search
PLAIN_MARGIN = 0.03        # a block pair must beat the allocator's own placement by this much to be worth its blocks
# A device that is (nearly) all ours can hand out one 96 GiB class contiguously: sixteen probes over 62 GiB of blocks and gaps,
# sixteen times 1.0 (the driver's box, round 4: the headline then ran at 0.68 of the peak instead of 0.87).  When the capped
# search ends without a clean pair and at least FAR_MIN_FREE_BYTES are free, candidates FAR behind the first block are tried:
# a gap of 64 / 96 / 128 GiB is allocated, the candidate block behind it, and the gap freed again at once -- held for the
# duration of two hipMalloc calls, never while probing.
FAR_GAPS_BYTES = (64 * GIB, 96 * GIB, 128 * GIB)
FAR_MIN_FREE_BYTES = 160 * GIB


class PlacementUnavailable(RuntimeError):
    """The device cannot provide the two arrays' own blocks: the caller allocates as it otherwise would."""


class DeviceBlock:
    """One ``gbl_block_alloc`` block (a plain hipMalloc on `device`), handed to torch through
    ``__cuda_array_interface__``: ``tensor()`` is a uint8 view of it, and the block is freed (``gbl_block_free``) when
    the last tensor over it and this object are gone."""

    def __init__(self, nbytes: int, device):
        self.nbytes, self.device, self.ptr = int(nbytes), torch.device(device), None
        free_parked()
        p = C.c_void_p()
        with torch.cuda.device(self.device):
            nat.check(nat.lib().gbl_block_alloc(self.nbytes, C.byref(p)), "gbl_block_alloc")
        self.ptr = p.value
        self.__cuda_array_interface__ = {"shape": (self.nbytes,), "typestr": "|u1", "data": (self.ptr, False),
                                         "version": 2, "strides": None}

    def tensor(self) -> torch.Tensor:
        return torch.as_tensor(self, device=self.device)

    def __del__(self):
        if getattr(self, "ptr", None):
            try:
                # hipFree synchronises the device, which is illegal while a stream capture is under way (a buffer dict
                # garbage-collected inside torch.cuda.graph): the block is then parked and freed with the next one
                if self.device.type == "cuda" and torch.cuda.is_current_stream_capturing():
                    _parked.append(self.ptr)
                else:
                    nat.lib().gbl_block_free(self.ptr)
                    free_parked()
            except Exception:  # noqa: BLE001  (interpreter shutdown)
                pass
            self.ptr = None


_parked: list[int] = []  # blocks whose last owner went away during a stream capture


def free_parked() -> None:
    """Free the blocks that could not be freed when their last owner went away (a stream capture was under way)."""
    while _parked and not torch.cuda.is_current_stream_capturing():
        nat.lib().gbl_block_free(_parked.pop())


def device_alloc(device):
    """The allocator ``spread_pair`` uses on a GPU: nbytes -> uint8 tensor over a block of its own."""
    return lambda nbytes: DeviceBlock(nbytes, device).tensor()


def free_bytes(device) -> int:
    f = C.c_int64()
    with torch.cuda.device(device):
        nat.check(nat.lib().gbl_device_memory(C.byref(f), None), "gbl_device_memory")
    return f.value


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


def block_bytes(nbytes: int) -> int:
    """The block an array of nbytes is carved from: whole 2 MiB pages, at least MIN_BLOCK_BYTES."""
    return max(MIN_BLOCK_BYTES, -(-int(nbytes) // BLOCK_GRANULE) * BLOCK_GRANULE)


_OOM = (torch.OutOfMemoryError, nat.GobbletHipError, MemoryError)


def spread_pair(bytes_a: int, bytes_b: int, device, slot_boards: int = 0, plies: int = 0, max_probes: int = MAX_PROBES,
                max_hold_bytes: int | None = None, alloc=None, free=None, plain=None, far: bool | None = None):
    """Two zero-filled uint8 tensors of bytes_a / bytes_b bytes on `device`, placed so that writes to them overlap.
    Returns (a, b, info); info records every probe, what was held and why the search ended.

    plain: () -> (a, b), the two arrays as the caller would otherwise allocate them (torch's allocator).  That pair is
    probed FIRST: if it is clean already it is used as it is -- no block, nothing pinned; otherwise the block search runs
    (each array the head of a block of its own, see the module docstring) and the plain pair stays in the race: the best
    pair seen wins, so the result is never worse than the caller's own placement (with most of a device taken the capped
    search may find nothing better).  alloc(nbytes) -> uint8 tensor (raising on out-of-memory) and free() -> free bytes:
    the allocator and the memory gauge (tests script them).  far: a capped search that found nothing goes on with candidates
    behind transient gaps of 64 / 96 / 128 GiB (see FAR_GAPS_BYTES) -- None: only on a device that is mostly free
    (FAR_MIN_FREE_BYTES), True: whenever a gap fits beside the reserve (the caller owns the device), False: never."""
    t0 = time.perf_counter()
    dev = torch.device(device)
    alloc = alloc or device_alloc(dev)
    free = free or (lambda: free_bytes(dev))
    size = {"a": int(bytes_a), "b": int(bytes_b)}
    first = None  # (a, b, ratio) of the caller's own placement
    if plain is not None:
        pa, pb = plain()
        us_both, us_a, us_b = probe(pa, pb, slot_boards, plies)
        first = (pa, pb, us_both / max(us_a + us_b, 1e-9))
        if first[2] <= ACCEPT_RATIO:
            pa.zero_(); pb.zero_()
            return pa, pb, {"spread": True, "ratio": round(first[2], 3), "probes": [round(first[2], 3)], "block_gib": [0.0, 0.0],
                            "held_gib": 0.0, "cap_gib": 0.0, "released_blocks": 0,
                            "ended": "the allocator's own placement is clean", "seconds": round(time.perf_counter() - t0, 3)}

    def keep_plain(why):
        pa, pb, r = first
        pa.zero_(); pb.zero_()
        return pa, pb, {"spread": bool(r <= SPREAD_RATIO), "ratio": round(r, 3), "probes": [round(r, 3)], "block_gib": [0.0, 0.0],
                        "held_gib": 0.0, "cap_gib": 0.0, "released_blocks": 0, "ended": why,
                        "seconds": round(time.perf_counter() - t0, 3)}

    block = {k: block_bytes(v) for k, v in size.items()}
    own = block["a"] + block["b"]
    free0 = free()
    if free0 < own + RESERVE_BYTES:
        why = ("%.1f GiB free, the arrays' own blocks need %.1f GiB + %d GiB reserve"
               % (free0 / GIB, own / GIB, RESERVE_BYTES // GIB))
        if first is not None:
            return keep_plain("no block search: " + why)
        raise PlacementUnavailable(why)
    cap = min(MAX_HOLD_BYTES if max_hold_bytes is None else int(max_hold_bytes), free0 // FREE_FRACTION)
    cap = max(cap, own)
    try:
        pool = {"a": [alloc(block["a"])]}
        pool["b"] = [alloc(block["b"])]
    except _OOM as e:
        if first is not None:
            return keep_plain("no block search: the device refused the arrays' own blocks")
        raise PlacementUnavailable("the device refused the arrays' own blocks: %s" % e) from None
    held = own
    tried = []
    best = (None, 0, 0)  # ratio, index into pool a, index into pool b

    def try_pair(ia, ib):
        nonlocal best
        us_both, us_a, us_b = probe(pool["a"][ia][:size["a"]], pool["b"][ib][:size["b"]], slot_boards, plies)
        ratio = us_both / max(us_a + us_b, 1e-9)
        tried.append(round(ratio, 3))
        if best[0] is None or ratio < best[0]:
            best = (ratio, ia, ib)

    try_pair(0, 0)
    grow = "b"  # blocks are added alternately: one for the mask array first (the smaller one)
    skips, skip = [], MIN_BLOCK_BYTES
    ended = "clean pair"
    while best[0] > ACCEPT_RATIO:
        if len(tried) >= max_probes:
            ended = "probe budget"
            break
        room = free()
        if held + block[grow] > cap or room < block[grow] + RESERVE_BYTES:
            ended = "memory budget (%.0f GiB: 1/%d of the %.0f GiB free at entry, at most %d)" % (
                cap / GIB, FREE_FRACTION, free0 / GIB, MAX_HOLD_BYTES // GIB)
            break
        try:
            # On a fresh device consecutive blocks can stay inside one class for tens of GiB (a class is 96 GiB): after a
            # plain conflict leave a gap first, twice as large each time
            if tried[-1] > SAME_RATIO and held + skip + block[grow] <= cap and room >= skip + block[grow] + RESERVE_BYTES:
                skips.append(alloc(skip))
                held += skip
                skip = min(2 * skip, MAX_SKIP_BYTES)
            pool[grow].append(alloc(block[grow]))
        except _OOM:
            ended = "the device refused a block"
            break
        held += block[grow]
        new, other = len(pool[grow]) - 1, "b" if grow == "a" else "a"
        # against the other array's first block; against the rest only if that pair is neither clean nor a plain
        # conflict (an array that straddles two classes), since all members of a pool probed alike so far
        for k in range(len(pool[other])):
            if best[0] <= ACCEPT_RATIO or len(tried) >= max_probes or (k > 0 and tried[-1] > SAME_RATIO):
                break
            try_pair(*((new, k) if grow == "a" else (k, new)))
        grow = other
    far_gaps, transient = [], 0
    if far is not False and best[0] > ACCEPT_RATIO and ended != "the device refused a block":
        # the blocks and gaps that led nowhere go back first (all but the arrays' first blocks and the best pair so far)
        keep_a, keep_b = {0, best[1]}, {0, best[2]}
        for k in range(len(pool["a"])):
            if k not in keep_a:
                pool["a"][k] = None
        for k in range(len(pool["b"])):
            if k not in keep_b:
                pool["b"][k] = None
        skips.clear()
        peak = held
        held = sum(block["a"] for x in pool["a"] if x is not None) + sum(block["b"] for x in pool["b"] if x is not None)
        for gap in FAR_GAPS_BYTES:
            if free() < max(0 if far else FAR_MIN_FREE_BYTES, gap + block["b"] + RESERVE_BYTES):
                break
            try:
                hold = alloc(gap)
                transient = max(transient, held + gap + block["b"])
                blk = alloc(block["b"])
            except _OOM:
                ended += "; the device refused a far candidate"
                break
            finally:
                hold = None  # (the gap goes back to the driver before anything is probed)
            pool["b"].append(blk)
            held += block["b"]
            far_gaps.append(gap // GIB)
            try_pair(0, len(pool["b"]) - 1)
            if best[0] <= ACCEPT_RATIO:
                ended = "clean pair behind a transient gap of %d GiB" % (gap // GIB)
                break
        held = max(held, peak)  # (what is reported: the most that was held while a probe ran)
    ratio, ia, ib = best
    if first is not None and first[2] <= ratio + PLAIN_MARGIN:  # nothing (clearly) better than the caller's own placement turned up
        out = keep_plain("%s; the allocator's own placement (%.3f) was not beaten" % (ended, first[2]))
        out[2].update(probes=[round(first[2], 3)] + tried, held_gib=round(held / GIB, 1), cap_gib=round(cap / GIB, 1),
                      released_blocks=sum(x is not None for x in pool["a"]) + sum(x is not None for x in pool["b"]) + len(skips),
                      far_gaps_gib=far_gaps, transient_peak_gib=round(transient / GIB, 1))
        pool.clear(); skips.clear()
        return out
    if first is not None:
        tried.insert(0, round(first[2], 3))
    a, b = pool["a"][ia][:size["a"]], pool["b"][ib][:size["b"]]
    a.zero_(); b.zero_()  # (a probe writes only zeros, but say so explicitly)
    # the caller's own pair lost: its memory goes back to torch's caching allocator, which keeps it for the process (it is not
    # returned to the driver unless somebody calls torch.cuda.empty_cache()) -- recorded, so that the footprint is not a surprise
    stranded = (size["a"] + size["b"]) / GIB if first is not None else 0.0
    released = sum(x is not None for x in pool["a"]) + sum(x is not None for x in pool["b"]) - 2 + len(skips)
    pool.clear()    # the rejected blocks and the gaps go back to the driver here (hipFree, no allocator cache involved)
    skips.clear()
    return a, b, {"spread": bool(ratio <= SPREAD_RATIO), "ratio": round(ratio, 3), "probes": tried,
                  "block_gib": [round(block["a"] / GIB, 3), round(block["b"] / GIB, 3)], "held_gib": round(held / GIB, 1),
                  "cap_gib": round(cap / GIB, 1), "released_blocks": released, "ended": ended,
                  "torch_cache_gib": round(stranded, 3), "far_gaps_gib": far_gaps, "transient_peak_gib": round(transient / GIB, 1),
                  "seconds": round(time.perf_counter() - t0, 3)}
