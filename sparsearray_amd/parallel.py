"""Multi-GPU sharding of the hot path.

One process per GPU, `torch.distributed` (backend "nccl" = RCCL over xGMI on
the GPU box, "gloo" in the CPU tests).  The rule: split the largest operand,
move the smallest tensor.

* crossprod(A, Y) with a tall dense Y: shard the CONTRACTED dimension (rows).
  Rank g owns rows [r0, r1) of A (row-filtered leaves) and of Y, computes a
  full ncol x K partial, and the partials are all-reduced (ncol*K doubles).
* col stats / rowsum: shard leaves (columns) by nnz; results are gathered.

The local compute is passed in as a callable so the same sharding logic runs
on the HIP path (bench.py) and under the CPU tests.
"""
from __future__ import annotations

from typing import Callable, Tuple

import numpy as np
import torch
import torch.distributed as dist


def row_block(nrow: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous row block of `rank` (sizes differ by at most one)."""
    base, rem = divmod(nrow, world)
    r0 = rank * base + min(rank, rem)
    return r0, r0 + base + (1 if rank < rem else 0)


def row_shard_csc(col_ptr, row_idx, val, r0: int, r1: int):
    """Leaves restricted to rows [r0, r1), offsets rebased to r0.  Offsets are
    ascending inside a leaf (src/leaf_utils.h:12-15) so this is a filter that
    keeps order.  Works on numpy arrays or torch tensors (any device)."""
    if isinstance(row_idx, torch.Tensor):
        keep = (row_idx >= r0) & (row_idx < r1)
        csum = torch.zeros(row_idx.numel() + 1, dtype=torch.int64, device=row_idx.device)
        csum[1:] = torch.cumsum(keep.to(torch.int64), 0)
        new_ptr = csum[col_ptr]
        return new_ptr, (row_idx[keep] - r0).to(torch.int32), val[keep]
    keep = (row_idx >= r0) & (row_idx < r1)
    csum = np.zeros(len(row_idx) + 1, dtype=np.int64)
    csum[1:] = np.cumsum(keep)
    return csum[np.asarray(col_ptr)], (row_idx[keep] - r0).astype(np.int32), val[keep]


def col_blocks_by_nnz(col_ptr, world: int):
    """Leaf ranges [c0, c1) per rank with ~equal nnz (prefix sums of col_ptr)."""
    cp = col_ptr.cpu().numpy() if isinstance(col_ptr, torch.Tensor) else np.asarray(col_ptr)
    ncol, nnz = len(cp) - 1, int(cp[-1])
    cuts = [0]
    for g in range(1, world):
        cuts.append(int(np.searchsorted(cp, nnz * g / world, side="left")))
    cuts.append(ncol)
    cuts = np.maximum.accumulate(np.clip(cuts, 0, ncol))
    return [(int(cuts[g]), int(cuts[g + 1])) for g in range(world)]


def sharded_crossprod(local_crossprod: Callable[[], torch.Tensor], group=None) -> torch.Tensor:
    """`local_crossprod()` returns this rank's ncol x K partial (any layout, the
    same on every rank); the sum over ranks is returned on every rank."""
    part = local_crossprod()
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(part, op=dist.ReduceOp.SUM, group=group)
    return part


def gather_columns(local: torch.Tensor, sizes, group=None) -> torch.Tensor:
    """Concatenate per-rank result slices (col stats: one scalar per leaf)."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return local
    outs = [torch.empty(n, dtype=local.dtype, device=local.device) for n in sizes]
    dist.all_gather(outs, local, group=group) if len(set(sizes)) == 1 else \
        _all_gather_ragged(outs, local, group)
    return torch.cat(outs)


def _all_gather_ragged(outs, local, group):
    world = dist.get_world_size(group)
    for src in range(world):
        buf = outs[src]
        if dist.get_rank(group) == src:
            buf.copy_(local)
        dist.broadcast(buf, src=src, group=group)
