#!/usr/bin/env python3
"""Time decode variants on one GPU (development aid, not the headline bench).
    python tools/bench_variants.py [--size 8192] [--reps 20]"""
import argparse, ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import jpeg_amd as J
from jpeg_amd import _lib, synth

ap = argparse.ArgumentParser(); ap.add_argument("--size", type=int, default=8192); ap.add_argument("--reps", type=int, default=20); ap.add_argument("--only", default=""); ap.add_argument("--l2", action="store_true", help="256 images of 512x512 all aliasing ONE input/output image: no HBM traffic, compute-bound time")
args = ap.parse_args()
ctx = J.Context(0); dev = ctx.torch_device; lib = _lib.lib()
q_np = np.stack([J.compression_quanta("luminance", 1.0), J.compression_quanta("chrominance", 1.0)])
d_q = torch.from_numpy(q_np.view(np.int16)).to(dev)
W = H = args.size
RING = 4
NIMG = 1
if args.l2:
    W = H = 512; NIMG = (args.size // 512) ** 2; RING = 1

def run(name, comps, fmt, color):
    if args.only and args.only not in name: return
    layout = J.Layout(fmt, comps)
    units = layout.units((W, H))
    q = [0] + [1] * (len(units) - 1)
    L = layout.c_layout((W, H), units, q)
    planes = synth.natural_planes_torch(units, RING, dev, 3)
    out = torch.empty((RING, W * H * 3), dtype=torch.uint8, device=dev)
    strides = _lib.size_array([0 if args.l2 else 64 * a * b for a, b in units])
    def step(i):
        r = i % RING
        st = lib.jpeg_amd_decode_batch(ctx.handle, C.byref(L), NIMG, _lib.ptr_array([p[r].data_ptr() for p in planes]), strides,
                                       d_q.data_ptr(), 0, 2, 0, color, out[r].data_ptr(), 0 if args.l2 else W * H * 3)
        assert st == 0, st
    for i in range(3): step(i)
    torch.cuda.synchronize()
    ctx.timer_begin()
    for i in range(args.reps): step(i)
    ms = ctx.timer_end() / args.reps
    nbytes = (128 * sum(a * b for a, b in units) + 3 * W * H) * NIMG
    print(f"{name:28s} {ms*1e3:8.1f} us  {W*H*NIMG/ms/1e3:10.0f} Mpx/s  {nbytes/ms/1e6:8.0f} GB/s alg  ({nbytes/1e6:.0f} MB)")
    del planes, out

run("grey -> RGB", {1: J.Component((1, 1), 0)}, "y8", _lib.COLOR_RGB8)
run("grey -> YCC", {1: J.Component((1, 1), 0)}, "y8", _lib.COLOR_YCC8)
c420 = {1: J.Component((2, 2), 0), 2: J.Component((1, 1), 1), 3: J.Component((1, 1), 1)}
c444 = {1: J.Component((1, 1), 0), 2: J.Component((1, 1), 1), 3: J.Component((1, 1), 1)}
c422 = {1: J.Component((2, 1), 0), 2: J.Component((1, 1), 1), 3: J.Component((1, 1), 1)}
run("4:2:0 -> RGB", c420, "ycc8", _lib.COLOR_RGB8)
run("4:2:0 -> YCC", c420, "ycc8", _lib.COLOR_YCC8)
run("4:2:2 -> RGB", c422, "ycc8", _lib.COLOR_RGB8)
run("4:4:4 -> RGB", c444, "ycc8", _lib.COLOR_RGB8)
c440 = {1: J.Component((1, 2), 0), 2: J.Component((1, 1), 1), 3: J.Component((1, 1), 1)}
run("4:4:0 -> RGB", c440, "ycc8", _lib.COLOR_RGB8)
