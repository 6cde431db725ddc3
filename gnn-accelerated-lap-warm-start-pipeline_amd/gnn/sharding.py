"""Batch sharding across the GPUs of one node.

Instances are independent (no cross-instance state anywhere on the hot path), so the batch is
cut into contiguous slices, every rank runs the whole pipeline on its slice, and the only
exchange is ONE gather of the assignments to rank 0 (RCCL over xGMI with the "nccl" backend;
"gloo" in the CPU tests).  Message per rank: slice x n x 8 bytes -- latency-bound, far below
the per-link xGMI bandwidth, so a single direct gather (not a ring) is the right collective."""
from __future__ import annotations

from typing import List, Optional, Tuple

import torch
import torch.distributed as dist


def shard_bounds(total: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous slice [lo, hi) of `total` instances owned by `rank`; sizes differ by at most 1."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError("bad world/rank")
    base, extra = divmod(total, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def gather_assignments(x_local: torch.Tensor, dst: int = 0, group=None,
                       sizes: Optional[List[int]] = None) -> Optional[torch.Tensor]:
    """Gather per-rank assignment blocks (rows = instances) on `dst`; returns the stacked tensor
    there and None elsewhere.  `sizes` = rows per rank when the slices are ragged."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    if sizes is None:
        sizes = [x_local.shape[0]] * world
    if len(set(sizes)) == 1:
        bufs = [torch.empty_like(x_local) for _ in range(world)] if rank == dst else None
        dist.gather(x_local.contiguous(), bufs, dst=dst, group=group)
        return torch.cat(bufs, dim=0) if rank == dst else None
    # ragged: pad to the largest slice so that one collective still suffices
    width = max(sizes)
    pad = torch.full((width,) + tuple(x_local.shape[1:]), -1, dtype=x_local.dtype, device=x_local.device)
    pad[: x_local.shape[0]] = x_local
    bufs = [torch.empty_like(pad) for _ in range(world)] if rank == dst else None
    dist.gather(pad, bufs, dst=dst, group=group)
    if rank != dst:
        return None
    return torch.cat([b[:s] for b, s in zip(bufs, sizes)], dim=0)
