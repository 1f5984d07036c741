"""Multi-GPU sharding of the hot path: one process per GPU, torch.distributed ("nccl" = RCCL
over xGMI on ROCm; "gloo" in CPU tests).

* MSM — point-range sharding: rank g keeps bases[g*n/N .. (g+1)*n/N) resident and receives the matching
  scalar slice; it computes a full partial MSM locally (no data-path collective).  The only exchange is an
  all-gather of the N partial results (96 B Jacobian each — RCCL has no elliptic-curve reduction op), after
  which every rank sums the N points on its own GPU (h2hip_g1_sum_jacobian_dev).
* NTT — independent columns are dealt round-robin to ranks (`columns_for_rank`); no communication.
"""
from __future__ import annotations

import numpy as np

from .h2hip import POINT_JACOBIAN, Bases, Context


def shard_range(n: int, rank: int, world: int):
    """[lo, hi) of the point range owned by `rank` (contiguous, sizes differ by at most one)."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def columns_for_rank(num_columns: int, rank: int, world: int):
    return list(range(rank, num_columns, world))


def sharded_msm(ctx: Context, bases: Bases, scalars_dptr: int, n_local: int, group=None, device=None, point_format: int = POINT_JACOBIAN) -> np.ndarray:
    """Partial MSM over this rank's slice, all-gather of the partials, local sum.  Every rank returns the full result."""
    import torch
    import torch.distributed as dist

    part = ctx.msm_dev(bases, scalars_dptr, n_local, POINT_JACOBIAN)
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        if point_format == POINT_JACOBIAN:
            return part
        t = torch.from_numpy(part.view(np.int64).copy()).reshape(1, 1, 12)
        t = t.to(device) if device is not None else t
        return _sum_gathered(ctx, t, 1, 1, point_format, device)
    world = dist.get_world_size(group)
    t = torch.from_numpy(part.view(np.int64).copy()).reshape(1, 12)
    if device is not None:
        t = t.to(device)
    gathered = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(gathered, t, group=group)
    allp = torch.cat(gathered, dim=0).contiguous()
    return _sum_gathered(ctx, allp, world, 1, point_format, device)[0:1]


def _sum_gathered(ctx: Context, allp, world: int, count: int, point_format: int, device) -> np.ndarray:
    """allp: (count, world, 12) int64 tensor of Jacobian partials -> (count, 8|12) sums computed on the GPU.
    A tensor gathered on the CPU (gloo) is staged into device memory first: the library only takes device pointers."""
    import torch

    width = 12 if point_format == POINT_JACOBIAN else 8
    out = np.zeros((count, width), dtype=np.uint64)
    staged = None
    if allp.device.type == "cpu":
        host = allp.numpy().view(np.uint64).reshape(-1)
        staged = ctx.to_device(host)
        base = staged
    else:
        torch.cuda.current_stream(device).synchronize()
        base = allp.data_ptr()
    try:
        stride = world * 12 * 8
        for j in range(count):
            out[j] = ctx.g1_sum_jacobian_dev(base + j * stride, world, point_format)[0]
    finally:
        if staged is not None:
            ctx.free(staged)
    return out


def sharded_msm_batch(ctx: Context, bases: Bases, scalar_dptrs, n_local: int, group=None, device=None) -> np.ndarray:
    """`len(scalar_dptrs)` independent MSMs (e.g. all columns of a phase) over this rank's slice, pipelined on two
    streams; ONE all-gather carries all the partials (count x 96 B per rank); every rank returns all full results
    as a (count, 12) Jacobian array."""
    import torch
    import torch.distributed as dist

    count = len(scalar_dptrs)
    parts = ctx.msm_batch_dev(bases, scalar_dptrs, n_local, POINT_JACOBIAN)
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return parts
    world = dist.get_world_size(group)
    t = torch.from_numpy(parts.view(np.int64).copy()).reshape(1, count, 12)
    if device is not None:
        t = t.to(device)
    gathered = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(gathered, t, group=group)
    allp = torch.cat(gathered, dim=0).permute(1, 0, 2).contiguous()   # (count, world, 12)
    return _sum_gathered(ctx, allp, world, count, POINT_JACOBIAN, device)
