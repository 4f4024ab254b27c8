"""Multi-GPU plumbing: one process per GPU, torch.distributed (backend "nccl" = RCCL over xGMI).

The spectral pipeline shards by independent images with NO data-path collective
(SURVEY.md section 8e).  The only collective is a broadcast of the quantisation tables from
rank 0 at batch start; timing is the max over ranks.  These helpers are backend-agnostic so
that the same logic is covered on CPU with gloo (tests/test_distributed_cpu.py).
"""
from __future__ import annotations

import numpy as np


def shard(n_items: int, rank: int, world: int):
    """Contiguous shard [lo, hi) of n_items for `rank` (chunks differ by at most one item)."""
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def broadcast_quanta(quanta, src: int, device, dist=None):
    """Broadcast uint16 tables [ntables, 64] from `src`; other ranks pass quanta=None plus the
    expected table count via `quanta` = int.  Returns an int16-typed tensor on `device` holding
    the uint16 bit patterns (what the kernels read)."""
    import torch
    if isinstance(quanta, (int, np.integer)):
        t = torch.zeros((int(quanta), 64), dtype=torch.int16, device=device)
    else:
        q = np.ascontiguousarray(np.asarray(quanta, np.uint16).reshape(-1, 64))
        t = torch.from_numpy(q.view(np.int16).copy()).to(device)
    if dist is not None and dist.is_initialized() and dist.get_world_size() > 1:
        # neither gloo nor NCCL/RCCL has a 16-bit integer type: ship the 128-byte tables as int32
        dist.broadcast(t.view(torch.int32), src=src)
    return t


def max_over_ranks(value: float, device, dist=None) -> float:
    import torch
    t = torch.tensor([value], dtype=torch.float64, device=device)
    if dist is not None and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
