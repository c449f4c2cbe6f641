"""Multi-GPU plumbing: series are independent GPs, so the batch dimension is sharded
contiguously across ranks -- one process per GPU, torch.distributed over RCCL ("nccl" backend on
ROCm; "gloo" in the CPU tests) -- and the ONLY collective on the path is an all-reduce of a few
scalars per step (summed loss; gradients of parameters shared across series)."""
from __future__ import annotations

import torch


def _dist():
    import torch.distributed as dist
    return dist if dist.is_available() and dist.is_initialized() else None


def shard_range(total: int, rank: int | None = None, world: int | None = None):
    """Contiguous [lo, hi) slice of `total` series owned by `rank` (remainder to the low ranks)."""
    d = _dist()
    if rank is None:
        rank = d.get_rank() if d else 0
    if world is None:
        world = d.get_world_size() if d else 1
    base, rem = divmod(total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def all_reduce_scalars(t: torch.Tensor, group=None) -> torch.Tensor:
    """Sum a small tensor of scalars over ranks (returns `t` unchanged when not distributed)."""
    d = _dist()
    if d is None or d.get_world_size(group) == 1:
        return t
    t = t.clone()
    d.all_reduce(t, op=d.ReduceOp.SUM, group=group)
    return t


def all_reduce_(t: torch.Tensor, group=None) -> torch.Tensor:
    d = _dist()
    if d is not None and d.get_world_size(group) > 1:
        d.all_reduce(t, op=d.ReduceOp.SUM, group=group)
    return t


def gather_samples(samples: torch.Tensor, group=None):
    """Optional final gather of per-series rollout samples [b_local, S, H] (the one place xGMI
    bandwidth is exercised).  Returns the list of per-rank tensors on every rank."""
    d = _dist()
    if d is None or d.get_world_size(group) == 1:
        return [samples]
    sizes = [None] * d.get_world_size(group)
    d.all_gather_object(sizes, tuple(samples.shape), group=group)
    nmax = max(s[0] for s in sizes)                       # all_gather wants equal shapes: pad the series dim
    padded = torch.zeros((nmax,) + tuple(samples.shape[1:]), dtype=samples.dtype, device=samples.device)
    padded[: samples.shape[0]] = samples
    outs = [torch.empty_like(padded) for _ in sizes]
    d.all_gather(outs, padded, group=group)
    return [o[: s[0]] for o, s in zip(outs, sizes)]
