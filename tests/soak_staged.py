#!/usr/bin/env python3
"""Soak of the GENERAL (staged) kernels -- the f-4 row: random sampling factors 1..4, 1..4 planes,
precision 8 / 12 / 16, centred and cosited upsampling, against the oracle.
    python tests/soak_staged.py <seed> <cases>   (not collected by pytest; uses the oracle)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import jpeg_amd as J
from oracle import oracle as O
ctx = J.Context(0)
rng = np.random.default_rng(int(sys.argv[1]))
N = int(sys.argv[2])
bad = 0
for it in range(N):
    n = int(rng.integers(1, 5))
    precision = int(rng.choice([8, 12, 16]))
    w, h = int(rng.integers(1, 300)), int(rng.integers(1, 200))
    comps = {i + 1: J.Component((int(rng.integers(1, 5)), int(rng.integers(1, 5))), int(rng.integers(0, 2))) for i in range(n)}
    layout = J.Layout(("custom", precision, n), comps)
    units = layout.units((w, h))
    amp = 1 << (precision + 1)
    planes = []
    for ux, uy in units:
        c = rng.integers(-amp, amp, (uy, ux, 64)).astype(np.int32)
        c[..., 6:] //= 16
        planes.append(np.clip(c, -32768, 32767).astype(np.int16))
    quanta = [rng.integers(1, 50, 64).astype(np.uint16) for _ in range(2)]
    q = [c.qi for c in layout.planes]
    keys = sorted(set(q)); q = [keys.index(k) for k in q]; tables = [quanta[k] for k in keys]
    cosite = bool(rng.integers(2))
    spectral = J.Spectral.from_host(ctx, (w, h), layout, planes, tables, q=q)
    planar = spectral.idct()
    want_p = [O.idct_plane(p, tables[i], precision) for p, i in zip(planes, q)]
    ok = all((a == b).all() for a, b in zip(planar.host_planes(), want_p))
    factors = [c.factor for c in layout.planes]
    rect = planar.interleaved(cosite=cosite).host_values()
    want_r = O.interleave(want_p, factors, layout.scale, (w, h), cosited=cosite)
    ok = ok and (rect == want_r).all()
    ok = ok and (spectral.rectangular(cosite=cosite).host_values() == want_r).all()   # one call: fused where every factor is 1 | 2
    # and back: decomposed + fdct of the interleaved samples
    back = J.Rectangular.from_host(ctx, (w, h), layout, want_r).decomposed()
    want_d = O.decompose(want_r.reshape(h, w, n), (w, h), factors, layout.scale)
    ok = ok and all((a == b).all() for a, b in zip(back.host_planes(), want_d))
    sp2 = back.fdct({c.qi: quanta[c.qi] for c in layout.planes})
    want_f = [O.fdct_plane(p, quanta[c.qi], precision) for p, c in zip(want_d, layout.planes)]
    ok = ok and all((a == b).all() for a, b in zip(sp2.host_planes(), want_f))
    if not ok:
        bad += 1
        print("MISMATCH", w, h, n, precision, cosite, [c.factor for c in layout.planes], flush=True)
print("staged soak done", N, "cases, mismatches:", bad)
