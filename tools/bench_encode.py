#!/usr/bin/env python3
"""Time the fused encode kernel (development aid):  python tools/bench_encode.py [--size 4096] [--height H] [--reps 50]"""
import argparse, ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import jpeg_amd as J
from jpeg_amd import _lib, synth

ap = argparse.ArgumentParser(); ap.add_argument("--size", type=int, default=4096); ap.add_argument("--reps", type=int, default=50)
ap.add_argument("--only", default=""); ap.add_argument("--height", type=int, default=0)
args = ap.parse_args()
ctx = J.Context(0); dev = ctx.torch_device; lib = _lib.lib()
W = args.size; H = args.height or args.size
RING = 4
q_np = np.stack([J.compression_quanta("luminance", 1.0), J.compression_quanta("chrominance", 1.0)])
d_q = torch.from_numpy(q_np.view(np.int16)).to(dev)
g = torch.Generator(device=dev); g.manual_seed(4)
# smooth + noise frame, cheap torch version
yy, xx = torch.meshgrid(torch.arange(H, device=dev), torch.arange(W, device=dev), indexing="ij")
base = 128 + 60 * torch.sin(xx / 97.0) * torch.cos(yy / 61.0)
px = [(base[..., None] + torch.randint(-8, 9, (H, W, 3), device=dev, generator=g) + 20 * i).clamp(0, 255).to(torch.uint8).reshape(-1).contiguous()
      for i in range(RING)]

def run(name, comps, fmt):
    if args.only and args.only not in name: return
    layout = J.Layout(fmt, comps)
    units = layout.units((W, H))
    q = [0] + [1] * (len(units) - 1)
    L = layout.c_layout((W, H), units, q)
    coefs = [torch.empty((RING, 64 * a * b), dtype=torch.int16, device=dev) for a, b in units]
    def step(i):
        r = i % RING
        st = lib.jpeg_amd_encode_batch(ctx.handle, C.byref(L), 1, px[r].data_ptr(), 0, _lib.COLOR_RGB8, d_q.data_ptr(), 0, 2,
                                       _lib.ptr_array([c[r].data_ptr() for c in coefs]), _lib.size_array([0] * len(units)))
        assert st == 0, st
    for i in range(3): step(i)
    torch.cuda.synchronize()
    ctx.timer_begin()
    for i in range(args.reps): step(i)
    ms = ctx.timer_end() / args.reps
    nbytes = 3 * W * H + 128 * sum(a * b for a, b in units)
    print(f"{name:18s} {ms*1e3:8.1f} us  {W*H/ms/1e3:10.0f} Mpx/s  {nbytes/ms/1e6:8.0f} GB/s alg  ({nbytes/1e6:.0f} MB)")

run("RGB -> grey", {1: J.Component((1, 1), 0)}, "y8")
run("RGB -> 4:2:0", {1: J.Component((2, 2), 0), 2: J.Component((1, 1), 1), 3: J.Component((1, 1), 1)}, "ycc8")
run("RGB -> 4:2:2", {1: J.Component((2, 1), 0), 2: J.Component((1, 1), 1), 3: J.Component((1, 1), 1)}, "ycc8")
run("RGB -> 4:4:4", {1: J.Component((1, 1), 0), 2: J.Component((1, 1), 1), 3: J.Component((1, 1), 1)}, "ycc8")
run("RGB -> 4:4:0", {1: J.Component((1, 2), 0), 2: J.Component((1, 1), 1), 3: J.Component((1, 1), 1)}, "ycc8")
