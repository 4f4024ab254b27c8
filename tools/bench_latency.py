#!/usr/bin/env python3
"""Per-call cost of decoding ONE small image (development aid): host time per call and GPU time."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import jpeg_amd as J
from jpeg_amd import _lib, synth
ctx = J.Context(0); dev = ctx.torch_device; lib = _lib.lib()
q_np = np.stack([J.compression_quanta("luminance", 1.0), J.compression_quanta("chrominance", 1.0)])
d_q = torch.from_numpy(q_np.view(np.int16)).to(dev)
for W, H in [(1920, 1080), (640, 480), (256, 256)]:
    layout = J.Layout("ycc8", {1: J.Component((2, 2), 0), 2: J.Component((1, 1), 1), 3: J.Component((1, 1), 1)})
    units = layout.units((W, H)); L = layout.c_layout((W, H), units, [0, 1, 1])
    planes = synth.natural_planes_torch(units, 1, dev, 3)
    out = torch.empty(W * H * 3, dtype=torch.uint8, device=dev)
    strides = _lib.size_array([0, 0, 0])
    ptrs = _lib.ptr_array([p[0].data_ptr() for p in planes])
    qh = np.ascontiguousarray(q_np)
    def dev_tables():
        return lib.jpeg_amd_decode_batch(ctx.handle, C.byref(L), 1, ptrs, strides, d_q.data_ptr(), 0, 2, 0, _lib.COLOR_RGB8, out.data_ptr(), 0)
    def host_tables():
        return lib.jpeg_amd_decode(ctx.handle, C.byref(L), ptrs, qh.ctypes.data, 2, 0, _lib.COLOR_RGB8, out.data_ptr())
    for name, fn in (("tables in HBM ", dev_tables), ("tables on host", host_tables)):
        for _ in range(20): assert fn() == 0
        torch.cuda.synchronize()
        N = 2000
        t0 = time.perf_counter(); ctx.timer_begin()
        for _ in range(N): fn()
        gpu_ms = ctx.timer_end(); host = time.perf_counter() - t0
        t0 = time.perf_counter()
        for _ in range(200): fn(); torch.cuda.synchronize()
        rt = (time.perf_counter() - t0) / 200
        print(f"{W}x{H} {name}: {host/N*1e6:6.1f} us/call issue rate, {gpu_ms/N*1e3:6.1f} us/call on the GPU timeline, {rt*1e6:6.1f} us call+sync")
