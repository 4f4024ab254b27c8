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
    if dist is not None and dist.is_initialized():   # (also a world of one: the collective then runs on the one rank)
        # neither gloo nor NCCL/RCCL has a 16-bit integer type: ship the 128-byte tables as int32
        dist.broadcast(t.view(torch.int32), src=src)
    return t


def max_over_ranks(value: float, device, dist=None) -> float:
    import torch
    t = torch.tensor([value], dtype=torch.float64, device=device)
    if dist is not None and dist.is_initialized():
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def band(size, scale, rank: int, world: int):
    """One huge image over several GPUs ("within one huge image"): horizontal
    bands of whole MCU rows, no exchange step.  Upsampling couples a pixel row with exactly one
    chroma sample row above / below (decode.swift:4243-4257), so a rank decodes its MCU rows plus
    ONE halo MCU row on each inner side as an independent sub-image and discards the halo rows;
    the result is bit-identical to the band of the whole-image decode.

    size = (W, H) pixels, scale = Layout.scale (MCU = 8 * scale pixels per axis).  Returns None
    for a rank without rows, else a dict:
      mcu_rows  (m0, m1)   MCU rows of the sub-image to decode (halo included)
      height    pixel height of that sub-image (the image's own bottom edge for the last rows)
      skip      pixel rows to drop at the top of the sub-image's output
      rows      (y0, y1)   pixel rows of the whole image this rank owns
    Plane p's coefficient rows for the sub-image are units rows [m0 * fy_p, ...) of the
    whole-image plane (`band_units`): contiguous in the reference's layout, so no copy."""
    w, h = size
    mh = 8 * scale[1]
    n_mcu = (h + mh - 1) // mh
    lo, hi = shard(n_mcu, rank, world)
    if lo == hi:
        return None
    m0, m1 = max(lo - 1, 0), min(hi + 1, n_mcu)
    return {"mcu_rows": (m0, m1), "height": min(h, m1 * mh) - m0 * mh, "skip": (lo - m0) * mh,
            "rows": (lo * mh, min(h, hi * mh))}


def band_units(plan, factor_y: int, units_y: int):
    """Row range [u0, u1) of a plane's units (block rows) that the sub-image of `plan` covers:
    factor_y block rows per MCU row, clipped to the plane (whose last MCU row may be partial)."""
    m0, m1 = plan["mcu_rows"]
    return m0 * factor_y, min(m1 * factor_y, units_y)
