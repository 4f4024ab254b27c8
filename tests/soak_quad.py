#!/usr/bin/env python3
"""Soak of k_quad420 (the 4:2:0 stack walk): random image sizes -- odd ones, whole stacks, short last stacks, partial tile
columns, planes that end inside a wave's window, both strip shapes -- as single images and small batches, RGB and YCbCr,
extreme and sparse coefficients, against the oracle.
    python tests/soak_quad.py <seed> <cases>   (not collected by pytest; it uses the oracle, so it lives under tests/)"""
import sys, os, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import jpeg_amd as J
from jpeg_amd import _lib
from oracle import oracle as O
ctx = J.Context(0); dev = ctx.torch_device
rng = np.random.default_rng(int(sys.argv[1]))
layout = J.Layout("ycc8", {1: J.Component((2, 2), 0), 2: J.Component((1, 1), 1), 3: J.Component((1, 1), 1)})
bad = 0
for it in range(int(sys.argv[2])):
    kind = int(rng.integers(5))
    if kind == 0: w, h = 256 * int(rng.integers(1, 17)), 64 * int(rng.integers(1, 9))                   # whole stacks of 32 x 2 strips
    elif kind == 1: w, h = 128 * int(rng.integers(1, 25)), 64 * int(rng.integers(1, 9)) + 16 * int(rng.integers(0, 4))
    elif kind == 2: w, h = 16 * int(rng.integers(1, 200)), int(rng.integers(1, 900))                    # vector stores, any height
    elif kind == 3: w, h = int(rng.integers(1, 2600)), int(rng.integers(1, 600))                        # byte-wise store tail
    else: w, h = int(rng.choice([1920, 3840, 1280, 640])), int(rng.choice([1080, 2160, 720, 360]))
    n = int(rng.choice([1, 1, 2, 3, 7])) if w * h < 1 << 20 else 1
    units = layout.units((w, h))
    batch = []
    for _ in range(n):
        planes = []
        for ux, uy in units:
            c = rng.integers(-1024, 1024, (uy, ux, 64)).astype(np.int16)
            if rng.integers(2): c[..., 8:] //= 8
            if rng.integers(3) == 0: c[..., 1:] = 0
            planes.append(c)
        batch.append(planes)
    quanta = [rng.integers(1, 64, 64).astype(np.uint16), rng.integers(1, 64, 64).astype(np.uint16)]
    rgb = bool(rng.integers(2))
    d_planes = [torch.from_numpy(np.stack([b[p] for b in batch])).to(dev) for p in range(3)]
    d_q = torch.from_numpy(np.stack(quanta).view(np.int16)).to(dev)
    out = torch.zeros((n, w * h * 3), dtype=torch.uint8, device=dev)
    L = layout.c_layout((w, h), units, [0, 1, 1])
    st = _lib.lib().jpeg_amd_decode_batch(ctx.handle, C.byref(L), n, _lib.ptr_array([p.data_ptr() for p in d_planes]),
                                          _lib.size_array([64 * a * b for a, b in units]), d_q.data_ptr(), 0, 2, 0,
                                          _lib.COLOR_RGB8 if rgb else _lib.COLOR_YCC8, out.data_ptr(), w * h * 3)
    assert st == 0, st
    got = out.cpu().numpy()
    for i in range(n):
        _, rect = O.decode(batch[i], [quanta[0], quanta[1], quanta[1]], [(2, 2), (1, 1), (1, 1)], (w, h), threads=8)
        want = O.unpack_rgb8(rect, 3, threads=8) if rgb else O.unpack_ycc8(rect, 3)
        if not (got[i] == want.reshape(-1)).all():
            bad += 1
            print("MISMATCH", w, h, n, i, rgb, int((got[i] != want.reshape(-1)).sum()))
print(f"quad soak done {sys.argv[2]} cases, mismatches: {bad}")
sys.exit(1 if bad else 0)
