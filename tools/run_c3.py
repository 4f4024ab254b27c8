#!/usr/bin/env python3
"""One 8192 x 8192 4:2:0 image (config 3) decoded `reps` times from a ring of inputs; prints us per call.
Development aid for profiler runs: tools/run_c3.py [reps] [W H N]."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import jpeg_amd as J
from jpeg_amd import _lib, synth
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
W, H, N = (int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (8192, 8192, 1)
ctx = J.Context(0); dev = ctx.torch_device; lib = _lib.lib()
q_np = np.stack([J.compression_quanta("luminance", 1.0), J.compression_quanta("chrominance", 1.0)])
d_q = torch.from_numpy(q_np.view(np.int16)).to(dev)
layout = J.Layout("ycc8", {1: J.Component((2, 2), 0), 2: J.Component((1, 1), 1), 3: J.Component((1, 1), 1)})
units = layout.units((W, H)); L = layout.c_layout((W, H), units, [0, 1, 1])
ring = 4
planes = [synth.natural_planes_torch(units, N, dev, 3 + r) for r in range(ring)]
out = torch.zeros((ring, N * W * H * 3), dtype=torch.uint8, device=dev)
strides = _lib.size_array([64 * a * b for a, b in units])
def step(i):
    r = i % ring
    st = lib.jpeg_amd_decode_batch(ctx.handle, C.byref(L), N, _lib.ptr_array([p.data_ptr() for p in planes[r]]), strides,
                                   d_q.data_ptr(), 0, 2, 0, _lib.COLOR_RGB8, out[r].data_ptr(), W * H * 3)
    assert st == 0, st
for i in range(5): step(i)
torch.cuda.synchronize(); ctx.timer_begin()
for i in range(reps): step(i)
print(f"{N} x {W}x{H} RGB: {ctx.timer_end() / reps * 1e3:.1f} us per call")
