"""Multi-GPU sharding of the board batch: boards are independent, so rank r of W owns
the contiguous global range ``shard_bounds(total, W, r)`` on its own GPU with its own tensors and
stream.  There is NO collective on the step path; the only communication is an optional sum of
a few int64 tallies after a run (``reduce_counters``).  The sampler is keyed by the global board id
(``env_base``), so trajectories do not depend on W."""
from __future__ import annotations

import torch


def shard_bounds(total_boards: int, world_size: int, rank: int) -> tuple[int, int]:
    """(first global board, number of boards) of `rank`; sizes differ by at most one."""
    if not (0 <= rank < world_size):
        raise ValueError("rank out of range")
    base, extra = divmod(int(total_boards), int(world_size))
    start = rank * base + min(rank, extra)
    return start, base + (1 if rank < extra else 0)


def make_shard(total_boards: int, rank: int, world_size: int, device, **kwargs):
    """This rank's ``BatchedGobblet`` over its shard of a `total_boards` batch."""
    from .vector_env import BatchedGobblet
    start, count = shard_bounds(total_boards, world_size, rank)
    return BatchedGobblet(count, device, env_base=start, **kwargs)


def reduce_counters(counters: torch.Tensor, group=None) -> torch.Tensor:
    """Sum the (4,) int64 tallies over ranks -- outside any timed / step path."""
    import torch.distributed as dist
    out = counters.clone()
    if dist.is_available() and dist.is_initialized():
        dist.all_reduce(out, op=dist.ReduceOp.SUM, group=group)
    return out
