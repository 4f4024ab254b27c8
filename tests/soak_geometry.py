#!/usr/bin/env python3
"""Soak: N random frame geometries / layouts / colour targets through the fused decode and encode
kernels against the oracle (a longer run of tests/test_gpu_parity.py::test_fused_kernels_on_random_geometry).
    python tests/soak_geometry.py <seed> <cases>   (not collected by pytest; it uses the oracle, so it lives under tests/)"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import jpeg_amd as J
from oracle import oracle as O
ctx = J.Context(0)
rng = np.random.default_rng(int(sys.argv[1]))
bad = 0
N = int(sys.argv[2])
for it in range(N):
    w = int(rng.choice([rng.integers(1, 50), rng.integers(50, 1200), 16 * rng.integers(1, 70), 256 * rng.integers(1, 5) + rng.integers(-9, 10)]))
    h = int(rng.choice([rng.integers(1, 50), rng.integers(50, 500), 16 * rng.integers(1, 30) + rng.integers(-9, 10)]))
    w, h = max(w, 1), max(h, 1)
    mode = int(rng.integers(5))
    fac = [(1, 1), (2, 1), (1, 2), (2, 2), (1, 1)][mode]
    grey = mode == 4
    comps = [((1, 1), 0)] if grey else [(fac, 0), ((1, 1), 1), ((1, 1), 1)]
    keyed = {i + 1: J.Component(f, qi) for i, (f, qi) in enumerate(comps)}
    layout = J.Layout("y8" if grey else "ycc8", keyed)
    units = layout.units((w, h))
    planes = []
    for ux, uy in units:
        c = rng.integers(-1024, 1024, (uy, ux, 64)).astype(np.int16)
        if rng.integers(2): c[..., 8:] //= 8
        planes.append(c)
    nq = 1 if grey else 2
    quanta = [rng.integers(1, 60, 64).astype(np.uint16) for _ in range(nq)]
    q = [qi for _, qi in comps]
    factors = [c.factor for c in layout.planes]
    spectral = J.Spectral.from_host(ctx, (w, h), layout, planes, quanta, q=q)
    rgb = bool(rng.integers(2))
    color, unpack, pack = (J.RGB, O.unpack_rgb8, O.pack_rgb8) if rgb else (J.YCbCr, O.unpack_ycc8, O.pack_ycc8)
    got = spectral.decode(color).cpu().numpy()
    _, rect = O.decode(planes, [quanta[i] for i in q], factors, (w, h))
    want = unpack(rect, len(comps))
    ok = (got == want).all()
    qmap = {i: quanta[i] for i in set(q)}
    coef = J.Rectangular.encode(ctx, (w, h), layout, want, qmap, color).host_planes()
    planar = O.decompose(pack(want, len(comps)).reshape(h, w, len(comps)), (w, h), factors, layout.scale)
    enc = [O.fdct_plane(p, quanta[i]) for p, i in zip(planar, q)]
    ok2 = all((a == b).all() for a, b in zip(coef, enc))
    if not (ok and ok2):
        bad += 1
        print("MISMATCH", w, h, fac, grey, rgb, ok, ok2, flush=True)
print("soak done", N, "cases, mismatches:", bad)
