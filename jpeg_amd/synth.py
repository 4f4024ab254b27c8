"""Seeded synthetic inputs for the BASELINE.json configurations.

Two coefficient distributions over 8x8 blocks of quantised int16 coefficients (zigzag):
  U  uniform in [-2048, 2047] in all 64 slots (stresses clamp / rounding);
  N  "natural": DC uniform in [-1024, 1023]; AC_z = round(Laplace(b = 200 / (1 + z))),
     so high frequencies are mostly zero like a real entropy-decoded stream.
The numpy generators (PCG64, seed 20240807 by default) feed the parity tests; the torch
generators build the large device-resident batches of bench.py directly in HBM.
"""
from __future__ import annotations

import numpy as np

SEED = 20240807

# the reference's "quality 1.0" tables: CompressionLevel.luminance(1.0) / .chrominance(1.0)
# (encode.swift:294-304, 320-332), and all-ones.


def blocks_uniform(nblocks: int, seed: int = SEED) -> np.ndarray:
    rng = np.random.Generator(np.random.PCG64(seed))
    return rng.integers(-2048, 2048, (nblocks, 64), dtype=np.int16)


def blocks_natural(nblocks: int, seed: int = SEED) -> np.ndarray:
    rng = np.random.Generator(np.random.PCG64(seed))
    scale = (200.0 / (1.0 + np.arange(64))).astype(np.float32)
    ac = rng.laplace(0.0, 1.0, (nblocks, 64)).astype(np.float32) * scale
    out = np.rint(ac).astype(np.int16)
    out[:, 0] = rng.integers(-1024, 1024, nblocks, dtype=np.int16)
    return out


def natural_planes_torch(units, n_images: int, device, seed: int = SEED):
    """Distribution N generated in HBM: list of int16 tensors [n_images, uy, ux, 64]."""
    import torch
    g = torch.Generator(device=device)
    planes = []
    scale = (200.0 / (1.0 + torch.arange(64, device=device, dtype=torch.float32)))
    for p, (ux, uy) in enumerate(units):
        g.manual_seed(seed + 7919 * p)
        out = torch.empty((n_images, uy, ux, 64), dtype=torch.int16, device=device)
        # generate in groups of images that keep the float32 temporaries around 256 MiB
        group = max(1, (1 << 26) // max(1, uy * ux * 64))
        for i in range(0, n_images, group):
            m = min(group, n_images - i)
            u = torch.rand((m, uy, ux, 64), generator=g, device=device, dtype=torch.float32) - 0.5
            lap = -torch.sign(u) * torch.log1p(-2.0 * u.abs()).clamp_(min=-30.0)
            v = torch.round(lap * scale).clamp_(-2047, 2047).to(torch.int16)
            v[..., 0] = torch.randint(-1024, 1024, (m, uy, ux), generator=g, device=device,
                                      dtype=torch.int16)
            out[i:i + m] = v
            del u, lap, v
        planes.append(out)
    return planes


def smooth_rgb(width: int, height: int, seed: int = SEED) -> np.ndarray:
    """C4 input: low-frequency sinusoids + uniform noise +-8, clamped; uint8 [H*W, 3]."""
    rng = np.random.Generator(np.random.PCG64(seed))
    y, x = np.mgrid[0:height, 0:width].astype(np.float32)
    img = np.empty((height, width, 3), np.float32)
    for c in range(3):
        fx, fy, ph = rng.uniform(0.002, 0.02), rng.uniform(0.002, 0.02), rng.uniform(0, 6.28)
        img[..., c] = 128 + 90 * np.sin(fx * x + ph) * np.cos(fy * y + 0.5 * ph)
    img += rng.integers(-8, 9, img.shape).astype(np.float32)
    return np.clip(np.rint(img), 0, 255).astype(np.uint8).reshape(-1, 3)


def smooth_rgb_torch(width: int, height: int, n_images: int, device, seed: int = SEED):
    """Device-side variant of smooth_rgb: uint8 tensor [n_images, H*W, 3]."""
    import torch
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    y = torch.arange(height, device=device, dtype=torch.float32)[:, None]
    x = torch.arange(width, device=device, dtype=torch.float32)[None, :]
    out = torch.empty((n_images, height * width, 3), dtype=torch.uint8, device=device)
    for i in range(n_images):
        par = torch.rand((3, 3), generator=g, device=device)
        img = torch.empty((height, width, 3), device=device)
        for c in range(3):
            fx = 0.002 + 0.018 * par[c, 0]
            fy = 0.002 + 0.018 * par[c, 1]
            ph = 6.28 * par[c, 2]
            img[..., c] = 128 + 90 * torch.sin(fx * x + ph) * torch.cos(fy * y + 0.5 * ph)
        img += torch.randint(-8, 9, img.shape, generator=g, device=device).float()
        out[i] = img.round_().clamp_(0, 255).to(torch.uint8).view(-1, 3)
    return out
