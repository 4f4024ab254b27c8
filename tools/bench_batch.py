#!/usr/bin/env python3
"""Batch decode rate for a few frame sizes (how much do partial strips cost?)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import jpeg_amd as J
from jpeg_amd import _lib, synth
ctx = J.Context(0); dev = ctx.torch_device; lib = _lib.lib()
q_np = np.stack([J.compression_quanta("luminance", 1.0), J.compression_quanta("chrominance", 1.0)])
d_q = torch.from_numpy(q_np.view(np.int16)).to(dev)
N = 128
for W, H in [(1920, 1080), (1792, 1072), (2048, 1088), (1920, 1088), (1792, 1080)]:
    layout = J.Layout("ycc8", {1: J.Component((2, 2), 0), 2: J.Component((1, 1), 1), 3: J.Component((1, 1), 1)})
    units = layout.units((W, H)); L = layout.c_layout((W, H), units, [0, 1, 1])
    ring = 2
    planes = [torch.stack([synth.natural_planes_torch([u], N, dev, 3 + r)[0] for r in range(ring)]) for u in units]
    out = torch.empty((ring, N * W * H * 3), dtype=torch.uint8, device=dev)
    strides = _lib.size_array([64 * a * b for a, b in units])
    def step(i):
        r = i % ring
        st = lib.jpeg_amd_decode_batch(ctx.handle, C.byref(L), N, _lib.ptr_array([p[r].data_ptr() for p in planes]), strides,
                                       d_q.data_ptr(), 0, 2, 0, _lib.COLOR_RGB8, out[r].data_ptr(), W * H * 3)
        assert st == 0, st
    for i in range(3): step(i)
    torch.cuda.synchronize(); ctx.timer_begin()
    for i in range(20): step(i)
    ms = ctx.timer_end() / 20
    print(f"{N} x {W}x{H}: {ms*1e3:8.1f} us  {N*W*H/ms/1e3:9.0f} Mpx/s")
    del planes, out
