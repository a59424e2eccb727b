"""EOT-sample sharding across ranks.

One process per GPU, a permanent frozen backbone replica per rank.  Every rank
holds the full (small) optimiser state and the same sampled mask indices; rank
``r`` pushes samples ``[r*S/world, (r+1)*S/world)`` of every image through the
backbone.  Per step there is exactly one data-path collective — an
``all_reduce(SUM)`` of the (B,3,H,W) fp32 patch gradient (602 112 B per image at
224²) over RCCL/xGMI — plus an ``all_gather`` of the tiny per-sample loss slab
for the bookkeeping.  The functions below are backend-agnostic (``nccl`` = RCCL
on ROCm, ``gloo`` in the CPU tests).
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


def allreduce_sum_(t, pg):
    """In-place sum over ranks of the patch gradient — THE data-path collective."""
    if pg is not None:
        import torch.distributed as dist
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=pg)
    return t


def allreduce_max_(t, pg):
    if pg is not None:
        import torch.distributed as dist
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=pg)
    return t


def gather_columns(local, pg):
    """(B, S_local) per rank -> (B, S) with rank r's columns at [r*S_local, (r+1)*S_local)."""
    if pg is None:
        return local
    import torch.distributed as dist
    world = dist.get_world_size(pg)
    parts = [torch.empty_like(local) for _ in range(world)]
    dist.all_gather(parts, local.contiguous(), group=pg)
    return torch.cat(parts, dim=1)
