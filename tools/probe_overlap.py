#!/usr/bin/env python3
"""Can k_chroma_idct (K1) hide under k_luma_fused (K2)?  Runs K2-only and K1-only builds of the
library (tools/build_exp.sh skipk1 -DJA_X_SKIPK1 / skipk2 -DJA_X_SKIPK2) on two streams."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import jpeg_amd as J
from jpeg_amd import _lib, synth
HERE = os.path.dirname(os.path.abspath(__file__))

def load(path):
    L = C.CDLL(path)
    for name, (res, args) in _lib.SIGNATURES.items():
        fn = getattr(L, name); fn.restype, fn.argtypes = res, args
    return L
libA = load(os.path.join(HERE, "exp/libjpeg_amd_skipk1.so"))   # K2 only
libB = load(os.path.join(HERE, "exp/libjpeg_amd_skipk2.so"))   # K1 only
dev = torch.device("cuda", 0)
sA, sB = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
def ctx(lib, stream):
    h = C.c_void_p(); assert lib.jpeg_amd_ctx_create(0, C.c_void_p(stream.cuda_stream), 0, C.byref(h)) == 0; return h
hA, hB = ctx(libA, sA), ctx(libB, sB)
W = H = 8192
layout = J.Layout("ycc8", {1: J.Component((2, 2), 0), 2: J.Component((1, 1), 1), 3: J.Component((1, 1), 1)})
units = layout.units((W, H)); L = layout.c_layout((W, H), units, [0, 1, 1])
q_np = np.stack([J.compression_quanta("luminance", 1.0), J.compression_quanta("chrominance", 1.0)])
d_q = torch.from_numpy(q_np.view(np.int16)).to(dev)
RING = 4
planes = synth.natural_planes_torch(units, RING, dev, 3)
outA = torch.empty((RING, W * H * 3), dtype=torch.uint8, device=dev)
outB = torch.empty((1, W * H * 3), dtype=torch.uint8, device=dev)
strides = _lib.size_array([64 * a * b for a, b in units])
def step(lib, h, r, out):
    st = lib.jpeg_amd_decode_batch(h, C.byref(L), 1, _lib.ptr_array([p[r].data_ptr() for p in planes]), strides,
                                   d_q.data_ptr(), 0, 2, 0, _lib.COLOR_RGB8, out.data_ptr(), W * H * 3)
    assert st == 0, st
N = 100
def run(doA, doB):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    for i in range(N):
        if doA: step(libA, hA, i % RING, outA[i % RING])
        if doB: step(libB, hB, (i + 2) % RING, outB[0])
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / N * 1e6
for _ in range(2):
    a = run(True, False); b = run(False, True); ab = run(True, True)
    print(f"K2 only {a:7.1f} us   K1 only {b:7.1f} us   both streams {ab:7.1f} us   (sum {a+b:.1f})")
