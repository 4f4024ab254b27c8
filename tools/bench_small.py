#!/usr/bin/env python3
"""One image per call at small and odd sizes, every layout: the single-call latency of the fused decode (HIP events, ring of 4 inputs).
    python tools/bench_small.py [--reps 50]"""
import argparse, ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import jpeg_amd as J
from jpeg_amd import _lib, synth
ap = argparse.ArgumentParser(); ap.add_argument("--reps", type=int, default=50)
args = ap.parse_args()
ctx = J.Context(0); dev = ctx.torch_device; lib = _lib.lib()
q_np = np.stack([J.compression_quanta("luminance", 1.0), J.compression_quanta("chrominance", 1.0)])
d_q = torch.from_numpy(q_np.view(np.int16)).to(dev)
c = J.Component
LAYOUTS = [("4:2:0", {1: c((2, 2), 0), 2: c((1, 1), 1), 3: c((1, 1), 1)}, "ycc8"), ("4:4:4", {1: c((1, 1), 0), 2: c((1, 1), 1), 3: c((1, 1), 1)}, "ycc8"),
           ("4:2:2", {1: c((2, 1), 0), 2: c((1, 1), 1), 3: c((1, 1), 1)}, "ycc8"), ("4:4:0", {1: c((1, 2), 0), 2: c((1, 1), 1), 3: c((1, 1), 1)}, "ycc8"),
           ("grey ", {1: c((1, 1), 0)}, "y8")]
RING = 4
print("decode -> RGB8, one image per call:")
for W, H in ((319, 480), (1919, 1079), (1920, 1080), (4095, 4095), (4096, 4096)):
    for name, comps, fmt in LAYOUTS:
        layout = J.Layout(fmt, comps)
        units = layout.units((W, H)); L = layout.c_layout((W, H), units, [0] + [1] * (len(units) - 1))
        planes = synth.natural_planes_torch(units, RING, dev, 3)
        out = torch.empty((RING, W * H * 3), dtype=torch.uint8, device=dev)
        strides = _lib.size_array([64 * a * b for a, b in units])
        def step(i):
            r = i % RING
            st = lib.jpeg_amd_decode_batch(ctx.handle, C.byref(L), 1, _lib.ptr_array([p[r].data_ptr() for p in planes]), strides,
                                           d_q.data_ptr(), 0, 2, 0, _lib.COLOR_RGB8, out[r].data_ptr(), W * H * 3)
            assert st == 0, st
        for i in range(5): step(i)
        torch.cuda.synchronize(); ctx.timer_begin()
        for i in range(args.reps): step(i)
        ms = ctx.timer_end() / args.reps
        nbytes = 128 * sum(a * b for a, b in units) + 3 * W * H
        print(f" {W:5d} x {H:5d} {name} : decode {ms * 1e3:8.1f} us ({nbytes / ms / 1e6:6.0f} GB/s)")
        del planes, out
