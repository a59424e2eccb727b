"""EOT-sample sharding across ranks.

One process per GPU, a permanent frozen backbone replica per rank.  Every rank
holds the full (small) optimiser state and draws the same mask indices from
identical generator state (synchronised once, at ``HotLoop`` construction);
rank ``r`` pushes samples ``[r*S/world, (r+1)*S/world)`` of every image through
the backbone.  Per step there is exactly ONE collective — an ``all_reduce(SUM)``
over RCCL/xGMI of the (B,3,H,W) fp32 patch gradient (602 112 B per image at
224²) with every rank's (B, S/world) loss / prediction columns and a draw
checksum riding in zero-padded slots behind it (x + 0 == x: the sum is the
gather).  Every 100 steps the failure sweep adds a MAX-reduce of a (B, 2520)
bitmap.  The functions below are backend-agnostic (``nccl`` = RCCL on ROCm,
``gloo`` in the CPU tests).
"""
import numpy as np
import torch


def world_rank(pg):
    if pg is None:
        return 1, 0
    import torch.distributed as dist
    return dist.get_world_size(pg), dist.get_rank(pg)


def src_rank(pg):
    import torch.distributed as dist
    return dist.get_global_rank(pg, 0)


def shard_bounds(S, world, rank):
    """Contiguous slice of the S samples owned by ``rank`` (S divisible by world)."""
    if S % world != 0:
        raise ValueError("sampling_size (%d) must be divisible by the number of ranks (%d)" % (S, world))
    per = S // world
    return rank * per, (rank + 1) * per


def mask_bounds(n_mask, world, rank):
    """Slice of the mask universe a rank sweeps in collect_failure (last rank may be short)."""
    per = (n_mask + world - 1) // world
    return min(n_mask, rank * per), min(n_mask, (rank + 1) * per)


def broadcast_(t, pg):
    if pg is not None:
        import torch.distributed as dist
        dist.broadcast(t, src=src_rank(pg), group=pg)
    return t


def broadcast_object(obj, pg):
    """Rank 0's picklable ``obj`` on every rank (setup only: RNG state / seeds)."""
    if pg is None:
        return obj
    import torch.distributed as dist
    box = [obj]
    dist.broadcast_object_list(box, src=src_rank(pg), group=pg)
    return box[0]


def all_true(flag, pg):
    """AND of a bool over the ranks (setup only: feature agreement, never on the data path)."""
    if pg is None:
        return bool(flag)
    import torch.distributed as dist
    votes = [None] * dist.get_world_size(pg)
    dist.all_gather_object(votes, bool(flag), group=pg)
    return all(votes)


def allreduce_sum_(t, pg):
    """In-place sum over ranks of the patch gradient (+ loss slabs) — THE data-path collective."""
    if pg is not None:
        import torch.distributed as dist
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=pg)
    return t


def allreduce_max_(t, pg):
    if pg is not None:
        import torch.distributed as dist
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=pg)
    return t
