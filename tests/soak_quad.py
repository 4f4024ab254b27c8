#!/usr/bin/env python3
"""Soak of k_luma_fused's QUAD walk: random 4:2:0 images made of whole 256 x 64-pixel stacks, 2048 / 4096 / 6144 wide, one to
six stacks high, RGB and YCbCr, extreme and sparse coefficients, against the oracle.
    python tests/soak_quad.py <seed> <cases>   (not collected by pytest; it uses the oracle, so it lives under tests/)"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import jpeg_amd as J
from oracle import oracle as O
ctx = J.Context(0)
rng = np.random.default_rng(int(sys.argv[1]))
bad = 0
for it in range(int(sys.argv[2])):
    w = int(rng.choice([2048, 4096, 6144])); h = 64 * int(rng.integers(1, 7))
    layout = J.Layout("ycc8", {1: J.Component((2, 2), 0), 2: J.Component((1, 1), 1), 3: J.Component((1, 1), 1)})
    planes = []
    for ux, uy in layout.units((w, h)):
        c = rng.integers(-1024, 1024, (uy, ux, 64)).astype(np.int16)
        if rng.integers(2): c[..., 8:] //= 8
        if rng.integers(3) == 0: c[..., 1:] = 0
        planes.append(c)
    quanta = [rng.integers(1, 64, 64).astype(np.uint16), rng.integers(1, 64, 64).astype(np.uint16)]
    rgb = bool(rng.integers(2))
    got = J.Spectral.from_host(ctx, (w, h), layout, planes, quanta, q=[0, 1, 1]).decode(J.RGB if rgb else J.YCbCr).cpu().numpy()
    _, rect = O.decode(planes, [quanta[0], quanta[1], quanta[1]], [(2, 2), (1, 1), (1, 1)], (w, h), threads=8)
    want = (O.unpack_rgb8 if rgb else O.unpack_ycc8)(rect, 3, threads=8) if rgb else O.unpack_ycc8(rect, 3)
    if not (got == want).all():
        bad += 1
        print("MISMATCH", w, h, rgb, int((got != want).sum()))
print(f"quad soak done {sys.argv[2]} cases, mismatches: {bad}")
sys.exit(1 if bad else 0)
