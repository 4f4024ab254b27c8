#!/usr/bin/env python3
"""8192 x 8192 4:2:0 decodes issued alternately on TWO contexts (two streams): the tail of one launch and the head of the next
overlap (every launch fills the chip, so the second one's workgroups start as the first one's leave).  Prints us per call for
one stream and for two.  Development aid: tools/run_c3_two_streams.py [reps]"""
import sys, os, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import jpeg_amd as J
from jpeg_amd import _lib, synth
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
W = H = 8192
lib = _lib.lib()
ctxs = [J.Context(0, own_stream=True), J.Context(0, own_stream=True)]
dev = ctxs[0].torch_device
q_np = np.stack([J.compression_quanta("luminance", 1.0), J.compression_quanta("chrominance", 1.0)])
d_q = torch.from_numpy(q_np.view(np.int16)).to(dev)
layout = J.Layout("ycc8", {1: J.Component((2, 2), 0), 2: J.Component((1, 1), 1), 3: J.Component((1, 1), 1)})
units = layout.units((W, H)); L = layout.c_layout((W, H), units, [0, 1, 1])
ring = 8
planes = [synth.natural_planes_torch(units, 1, dev, 3 + r) for r in range(ring)]
out = torch.zeros((ring, W * H * 3), dtype=torch.uint8, device=dev)
strides = _lib.size_array([64 * a * b for a, b in units])
torch.cuda.synchronize()
def step(i, nctx):
    r = i % ring
    st = lib.jpeg_amd_decode_batch(ctxs[i % nctx].handle, C.byref(L), 1, _lib.ptr_array([p.data_ptr() for p in planes[r]]), strides,
                                   d_q.data_ptr(), 0, 2, 0, _lib.COLOR_RGB8, out[r].data_ptr(), W * H * 3)
    assert st == 0, st
for nctx in (1, 2, 1, 2):
    for i in range(20): step(i, nctx)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(reps): step(i, nctx)
    torch.cuda.synchronize()
    us = (time.perf_counter() - t0) / reps * 1e6
    print(f"{nctx} stream(s): {us:.1f} us per 8192x8192 decode = {402653184 / us / 1e3 / 8000 * 100:.1f} % of 8 TB/s")
