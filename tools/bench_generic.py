#!/usr/bin/env python3
"""The JPEG.Format plug-in path: layouts outside the built-in 8-bit fast paths -- 12-bit, four planes, cosited,
4:1:1 -- decoded to Rectangular (UInt16 samples, `idct().interleaved(cosite:)`) and encoded from it (`decomposed().fdct(quanta:)`).
Staged = one kernel per stage with Planar in HBM (k_idct_plane per plane + k_planar_to_pixels; k_decompose per plane + k_fdct_plane
per plane); fused = jpeg_amd_spectral_rectangular (k_generic_fused, round 5) where the layout qualifies.
Algorithmic bytes ("stops at Rectangular"): 128 B per coefficient block + 2 B per sample of the rectangular image.
    python tools/bench_generic.py [--size 8192] [--reps 10] [--only name]"""
import argparse, ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import jpeg_amd as J
from jpeg_amd import _lib, synth

ap = argparse.ArgumentParser()
ap.add_argument("--size", type=int, default=8192); ap.add_argument("--height", type=int, default=0)
ap.add_argument("--reps", type=int, default=10); ap.add_argument("--only", default="")
args = ap.parse_args()
ctx = J.Context(0); dev = ctx.torch_device; lib = _lib.lib()
W = args.size; H = args.height or args.size
q_np = np.stack([J.compression_quanta("luminance", 1.0), J.compression_quanta("chrominance", 1.0)]).astype(np.uint16)
has_fused = hasattr(lib, "jpeg_amd_spectral_rectangular")


def timed(fn, reps):
    for _ in range(2): fn()
    torch.cuda.synchronize(); ctx.timer_begin()
    for _ in range(reps): fn()
    return ctx.timer_end() / reps


def run(name, fmt, comps, cosite):
    if args.only and args.only not in name: return
    layout = J.Layout(fmt, comps)
    units = layout.units((W, H)); n = layout.count
    q = [0] + [1] * (n - 1)
    L = layout.c_layout((W, H), units, q)
    P = layout.precision
    planes = synth.natural_planes_torch(units, 1, dev, 3)
    coef = [p[0] if p.dim() > 1 else p for p in planes]
    if P > 8:   # spread the synthetic 8-bit-scaled coefficients over the format's range
        coef = [(c.to(torch.int32) * (1 << (P - 8))).clamp(-32768, 32767).to(torch.int16) for c in coef]
    spatial = [torch.empty(64 * a * b, dtype=torch.int16, device=dev) for a, b in units]
    rect = torch.empty(W * H * n, dtype=torch.int16, device=dev)
    back = [torch.empty(64 * a * b, dtype=torch.int16, device=dev) for a, b in units]
    cp, sp, bp = (_lib.ptr_array([t.data_ptr() for t in ts]) for ts in (coef, spatial, back))
    qptr = q_np.ctypes.data_as(C.c_void_p)

    def staged_decode():
        assert lib.jpeg_amd_spectral_idct(ctx.handle, C.byref(L), cp, qptr, 2, sp) == 0
        assert lib.jpeg_amd_planar_interleaved(ctx.handle, C.byref(L), sp, 1 if cosite else 0, rect.data_ptr()) == 0

    def fused_decode():
        st = lib.jpeg_amd_spectral_rectangular(ctx.handle, C.byref(L), cp, qptr, 2, 1 if cosite else 0, rect2.data_ptr())
        assert st == 0, st

    def staged_encode():
        assert lib.jpeg_amd_rectangular_decomposed(ctx.handle, C.byref(L), rect.data_ptr(), sp) == 0
        assert lib.jpeg_amd_planar_fdct(ctx.handle, C.byref(L), sp, qptr, 2, bp) == 0

    nblocks = sum(a * b for a, b in units)
    nbytes = 128 * nblocks + 2 * W * H * n
    ms = timed(staged_decode, args.reps)
    line = f"{name:34s} decode staged {ms*1e3:9.1f} us {nbytes/ms/1e6:7.0f} GB/s"
    if has_fused:
        rect2 = torch.empty_like(rect)
        st = lib.jpeg_amd_spectral_rectangular(ctx.handle, C.byref(L), cp, qptr, 2, 1 if cosite else 0, rect2.data_ptr())
        if st == 0:
            same = bool(torch.equal(rect, rect2))
            msf = timed(fused_decode, args.reps)
            line += f" | fused {msf*1e3:9.1f} us {nbytes/msf/1e6:7.0f} GB/s ({ms/msf:4.1f} x, {'== staged' if same else 'DIFFERS'})"
        else:
            line += f" | fused: not supported ({st})"
    mse = timed(staged_encode, args.reps)
    line += f" | encode staged {mse*1e3:9.1f} us {nbytes/mse/1e6:7.0f} GB/s"
    if hasattr(lib, "jpeg_amd_rectangular_spectral"):
        back2 = [torch.empty_like(b) for b in back]
        bp2 = _lib.ptr_array([t.data_ptr() for t in back2])
        def fused_encode():
            st = lib.jpeg_amd_rectangular_spectral(ctx.handle, C.byref(L), rect.data_ptr(), qptr, 2, bp2)
            assert st == 0, st
        fused_encode()
        same = all(bool(torch.equal(a, b)) for a, b in zip(back, back2))
        msf = timed(fused_encode, args.reps)
        line += f" | fused {msf*1e3:9.1f} us {nbytes/msf/1e6:7.0f} GB/s ({mse/msf:4.1f} x, {'== staged' if same else 'DIFFERS'})"
    line += f"  [{nbytes/1e6:.0f} MB alg]"
    print(line, flush=True)


c = J.Component
print(f"{W} x {H}; algorithmic bytes = 128 B per block + 2 B per rectangular sample")
run("4:2:0 12-bit, 3 planes", ("custom", 12, 3), {1: c((2, 2), 0), 2: c((1, 1), 1), 3: c((1, 1), 1)}, False)
run("rgba12 (2,2)x3 + (1,1), 4 planes", ("custom", 12, 4), {4: c((2, 2), 0), 5: c((2, 2), 0), 6: c((2, 2), 0), 7: c((1, 1), 1)}, False)
run("4:2:0 8-bit cosited -> Rect16", ("custom", 8, 3), {1: c((2, 2), 0), 2: c((1, 1), 1), 3: c((1, 1), 1)}, True)
run("4:2:0 8-bit centred -> Rect16", ("custom", 8, 3), {1: c((2, 2), 0), 2: c((1, 1), 1), 3: c((1, 1), 1)}, False)
run("4:4:4 16-bit, 3 planes", ("custom", 16, 3), {1: c((1, 1), 0), 2: c((1, 1), 1), 3: c((1, 1), 1)}, False)
run("4:2:2 12-bit, 3 planes", ("custom", 12, 3), {1: c((2, 1), 0), 2: c((1, 1), 1), 3: c((1, 1), 1)}, False)
run("4:1:1 8-bit (4,1) + (1,1) x 2", ("custom", 8, 3), {1: c((4, 1), 0), 2: c((1, 1), 1), 3: c((1, 1), 1)}, False)
run("4:1:0 12-bit (4,2) + (1,1) x 2", ("custom", 12, 3), {1: c((4, 2), 0), 2: c((1, 1), 1), 3: c((1, 1), 1)}, False)
run("factor 3: (3,3) + (1,1) x 2, 12-bit", ("custom", 12, 3), {1: c((3, 3), 0), 2: c((1, 1), 1), 3: c((1, 1), 1)}, False)
