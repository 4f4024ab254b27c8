#!/usr/bin/env python3
"""Fused 4:2:0 decode time for one image of several sizes and for batches, RGB and YCbCr targets
(development aid; A/B against a tools/build_exp.sh build through JPEG_AMD_LIBRARY)."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import jpeg_amd as J
from jpeg_amd import _lib, synth
ctx = J.Context(0); dev = ctx.torch_device; lib = _lib.lib()
q_np = np.stack([J.compression_quanta("luminance", 1.0), J.compression_quanta("chrominance", 1.0)])
d_q = torch.from_numpy(q_np.view(np.int16)).to(dev)
for (W, H, N) in [(8192, 8192, 1), (4096, 4096, 1), (2048, 2048, 1), (1024, 1024, 1), (512, 512, 1), (2048, 2048, 16), (512, 512, 256)]:
    layout = J.Layout("ycc8", {1: J.Component((2, 2), 0), 2: J.Component((1, 1), 1), 3: J.Component((1, 1), 1)})
    units = layout.units((W, H)); L = layout.c_layout((W, H), units, [0, 1, 1])
    ring = 3
    planes = [synth.natural_planes_torch(units, N, dev, 3 + r) for r in range(ring)]
    out = torch.zeros((ring, N * W * H * 3), dtype=torch.uint8, device=dev)
    strides = _lib.size_array([64 * a * b for a, b in units])
    res = []
    for color in (_lib.COLOR_RGB8, _lib.COLOR_YCC8):
        def step(i):
            r = i % ring
            st = lib.jpeg_amd_decode_batch(ctx.handle, C.byref(L), N, _lib.ptr_array([p.data_ptr() for p in planes[r]]), strides,
                                           d_q.data_ptr(), 0, 2, 0, color, out[r].data_ptr(), W * H * 3)
            assert st == 0, st
        for i in range(5): step(i)
        torch.cuda.synchronize(); ctx.timer_begin()
        reps = 50
        for i in range(reps): step(i)
        res.append(ctx.timer_end() / reps * 1e3)
    print(f"{N:4d} x {W}x{H}: RGB {res[0]:8.1f} us   YCC {res[1]:8.1f} us", flush=True)
