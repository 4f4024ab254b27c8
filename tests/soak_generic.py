#!/usr/bin/env python3
"""Soak of k_generic_fused (jpeg_amd_spectral_rectangular): random custom formats -- 1..4 planes, every factor 1 .. 4 (round 6:
the fused kernels take every plane that is an integer fraction of the scale, the staged kernels the rest inside the same call),
precision 1..16, centred / cosited, a non-recognised component setting the scale now and then -- on images of
several tiles (up to 900 x 500), extreme and sparse coefficients, against the oracle (the reference's literal formulas).
    python tests/soak_generic.py <seed> <cases>   (not collected by pytest; uses the oracle)"""
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
    precision = int(rng.choice([8, 12, 16, int(rng.integers(1, 17))]))
    kind = int(rng.integers(4))
    if kind == 0: w, h = int(rng.integers(1, 900)), int(rng.integers(1, 500))
    elif kind == 1: w, h = 128 * int(rng.integers(1, 6)), 64 * int(rng.integers(1, 6))       # whole tiles
    elif kind == 2: w, h = 128 * int(rng.integers(1, 5)) + int(rng.integers(-17, 18)), 64 * int(rng.integers(1, 5)) + int(rng.integers(-17, 18))
    else: w, h = int(rng.integers(1, 40)), int(rng.integers(1, 40))
    fmax = int(rng.choice([2, 2, 3, 4, 4]))   # factors up to 2 (the layouts of round 5), 3 or 4
    pick = (lambda: int(rng.choice([f for f in (1, 2, 3, 4) if f <= fmax])))
    comps = {i + 1: J.Component((pick(), pick()), int(rng.integers(0, 2))) for i in range(n)}
    if n > 1 and rng.integers(4) == 0:   # (a single plane below the scale does not cover the image: the reference traps)
        comps[99] = J.Component((fmax if rng.integers(2) else 2, fmax if rng.integers(2) else 2), 0)    # not recognised: takes part in the scale only (decode.swift:2181-2190)
    layout = J.Layout(("custom", precision, n), comps)
    units = layout.units((w, h))
    amp = 1 << (precision + 1)
    planes = []
    for ux, uy in units:
        c = rng.integers(-amp, amp, (uy, ux, 64)).astype(np.int32)
        if rng.integers(2): c[..., 6:] //= 16
        if rng.integers(4) == 0: c[..., 1:] = 0
        planes.append(np.clip(c, -32768, 32767).astype(np.int16))
    quanta = [rng.integers(1, 50, 64).astype(np.uint16) for _ in range(2)]
    q = [c.qi for c in layout.planes]
    keys = sorted(set(q)); q = [keys.index(k) for k in q]; tables = [quanta[k] for k in keys]
    cosite = bool(rng.integers(2))
    spectral = J.Spectral.from_host(ctx, (w, h), layout, planes, tables, q=q)
    want_p = [O.idct_plane(p, tables[i], precision) for p, i in zip(planes, q)]
    want = O.interleave(want_p, [c.factor for c in layout.planes], layout.scale, (w, h), cosited=cosite)
    got = spectral.rectangular(cosite=cosite).host_values()
    if not (got == want).all():
        bad += 1
        print("MISMATCH", w, h, n, precision, cosite, [c.factor for c in layout.planes], layout.scale, int((got != want).sum()), flush=True)
    # ... and back: the interleaved samples through the fused Rectangular -> Spectral (decomposed() + fdct(quanta:))
    flist = [c.factor for c in layout.planes]
    enc = J.Rectangular.from_host(ctx, (w, h), layout, want).spectral({c.qi: quanta[c.qi] for c in layout.planes}).host_planes()
    want_d = O.decompose(want.reshape(h, w, n), (w, h), flist, layout.scale)
    want_f = [O.fdct_plane(p, quanta[c.qi], precision) for p, c in zip(want_d, layout.planes)]
    if not all((a == b).all() for a, b in zip(enc, want_f)):
        bad += 1
        print("MISMATCH (encode)", w, h, n, precision, flist, layout.scale, [int((a != b).sum()) for a, b in zip(enc, want_f)], flush=True)
print("generic soak done", N, "cases, mismatches:", bad)
sys.exit(1 if bad else 0)
