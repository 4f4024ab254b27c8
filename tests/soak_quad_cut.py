#!/usr/bin/env python3
"""Soak of the mixed column cut and of the ticket walk of k_quad420 (quad_cut, dynamic walk: kernels_quad.hip): batches long enough to be cut into 32 x 2 strips
for the whole columns + one column of 16 x 4 strips, random sizes whose remainder column is 1 .. 16 blocks wide -- odd
widths (byte-wise store tail), partial remainder columns, short last stacks on either side of the seam -- against the oracle
on a sample of the batch.
    python tests/soak_quad_cut.py <seed> <cases>   (not collected by pytest; it uses the oracle, so it lives under tests/)"""
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
def stacks(cols, uy, by): return cols * ((-(-uy // by) + 3) // 4)
bad = cut = 0
for it in range(int(sys.argv[2])):
    whole, rest = int(rng.integers(1, 5)), int(rng.integers(1, 17))
    ux = 32 * whole + rest
    w = 8 * ux - int(rng.integers(0, 8)) * int(rng.integers(0, 2))          # sometimes not a multiple of 8 / 16
    h = int(rng.integers(8, 400))
    units = layout.units((w, h))
    ux, uy = units[0]
    wide, narrow = stacks(-(-ux // 32), uy, 2), stacks(-(-ux // 16), uy, 4)
    mixed = stacks(ux // 32, uy, 2) + stacks(1, uy, 4)
    n = -(-16 * 768 // min(wide, narrow)) + int(rng.integers(0, 3))           # just long enough for the cut to be considered ...
    if rng.integers(2): n = n * 2                                             # ... or long enough for the ticket walk in both launches
    if w * h * n > 400_000_000: continue
    cut += int(mixed < min(wide, narrow) and 0 < ux % 32 <= 16)
    pool = [np.clip(rng.laplace(0, 60, (4, b, a, 64)), -1024, 1023).astype(np.int16) for a, b in units]
    idx = rng.integers(0, 4, n)
    d_planes = [torch.from_numpy(p[idx]).to(dev) for p in pool]
    quanta = [rng.integers(1, 40, 64).astype(np.uint16), rng.integers(1, 40, 64).astype(np.uint16)]
    d_q = torch.from_numpy(np.stack(quanta).view(np.int16)).to(dev)
    rgb = bool(rng.integers(2))
    out = torch.zeros((n, w * h * 3), dtype=torch.uint8, device=dev)
    L = layout.c_layout((w, h), units, [0, 1, 1])
    st = _lib.lib().jpeg_amd_decode_batch(ctx.handle, C.byref(L), n, _lib.ptr_array([p.data_ptr() for p in d_planes]),
                                          _lib.size_array([64 * a * b for a, b in units]), d_q.data_ptr(), 0, 2, 0,
                                          _lib.COLOR_RGB8 if rgb else _lib.COLOR_YCC8, out.data_ptr(), w * h * 3)
    assert st == 0, st
    want = {}
    for k in range(4):
        _, rect = O.decode([p[k] for p in pool], [quanta[0], quanta[1], quanta[1]], [(2, 2), (1, 1), (1, 1)], (w, h), threads=8)
        want[k] = (O.unpack_rgb8(rect, 3, threads=8) if rgb else O.unpack_ycc8(rect, 3)).reshape(-1)
    got = out.cpu().numpy()
    for i in range(n):                                                        # EVERY image of the batch
        if not (got[i] == want[int(idx[i])]).all():
            bad += 1
            print("MISMATCH", w, h, n, i, rgb, int((got[i] != want[int(idx[i])]).sum()))
            break
print(f"quad cut soak done {sys.argv[2]} cases ({cut} took the mixed cut), mismatches: {bad}")
sys.exit(1 if bad else 0)
