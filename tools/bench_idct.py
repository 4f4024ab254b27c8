#!/usr/bin/env python3
"""Time the staged IDCT + dequantise kernel alone (config-2 shape):  python tools/bench_idct.py [--units 2048]"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import jpeg_amd as J
from jpeg_amd import _lib, synth
ap = argparse.ArgumentParser(); ap.add_argument("--units", type=int, default=2048); ap.add_argument("--reps", type=int, default=30)
ap.add_argument("--only-main", action="store_true", help="the 2^22-block plane only (profiler runs)")
args = ap.parse_args()
ctx = J.Context(0); dev = ctx.torch_device; lib = _lib.lib()
q = np.ascontiguousarray(J.compression_quanta("luminance", 1.0))
for ux, uy in ([(args.units, args.units)] if args.only_main else [(args.units, args.units), (512, 512), (313, 320)]):
    ring = 3
    coef = synth.natural_planes_torch([(ux, uy)], ring, dev, 99)[0]
    plane = torch.empty((ring, 64 * ux * uy), dtype=torch.int16, device=dev)
    def step(i):
        st = lib.jpeg_amd_idct_plane(ctx.handle, coef[i % ring].data_ptr(), ux, uy, q.ctypes.data, 8, plane[i % ring].data_ptr())
        assert st == 0
    for i in range(5): step(i)
    torch.cuda.synchronize(); ctx.timer_begin()
    for i in range(args.reps): step(i)
    ms = ctx.timer_end() / args.reps
    nb = ux * uy
    print(f"{ux}x{uy} blocks  {ms*1e3:8.1f} us  {nb/ms/1e6:7.2f} Gblocks/s  {256*nb/ms/1e6:7.0f} GB/s  ({256*nb/1e6:.0f} MB)")
    del coef, plane
