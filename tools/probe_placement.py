#!/usr/bin/env python3
"""Does the C3 step depend on WHERE its buffers lie relative to each other?  (round 5: bench.py measured 76.5 us with a ring of 4
image sets and 80-82 us with rings of 3 and 8 -- every step touches ONE set, so only the addresses differ.)
One arena; the luma / Cb / Cr coefficient planes and the pixel buffer of an 8192 x 8192 4:2:0 image are placed at chosen byte
offsets inside it and the fused decode is timed (HIP events, 60 calls after 5).
    python tools/probe_placement.py"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import jpeg_amd as J
from jpeg_amd import _lib
ctx = J.Context(0); dev = ctx.torch_device; lib = _lib.lib()
W = H = 8192
layout = J.Layout("ycc8", {1: J.Component((2, 2), 0), 2: J.Component((1, 1), 1), 3: J.Component((1, 1), 1)})
units = layout.units((W, H)); L = layout.c_layout((W, H), units, [0, 1, 1])
q_np = np.stack([J.compression_quanta("luminance", 1.0), J.compression_quanta("chrominance", 1.0)])
d_q = torch.from_numpy(q_np.view(np.int16)).to(dev)
MiB = 1 << 20
arena = torch.empty(3 * 1024 * MiB, dtype=torch.uint8, device=dev)
arena.view(torch.int16).random_(-40, 40)          # small coefficients everywhere (the kernel's time does not depend on the values)
base = arena.data_ptr()
base += (-base) % (2 * MiB)                         # 2 MiB aligned origin
szY, szC, szO = 128 * MiB, 32 * MiB, 192 * MiB
strides = _lib.size_array([0, 0, 0])

def run(oY, oB, oR, oO, reps=60):
    ptrs = _lib.ptr_array([base + oY, base + oB, base + oR])
    def step():
        st = lib.jpeg_amd_decode_batch(ctx.handle, C.byref(L), 1, ptrs, strides, d_q.data_ptr(), 0, 2, 0, _lib.COLOR_RGB8, base + oO, W * H * 3)
        assert st == 0, st
    for _ in range(5): step()
    torch.cuda.synchronize(); ctx.timer_begin()
    for _ in range(reps): step()
    return ctx.timer_end() / reps * 1e3

K = 1024
print("placement of (Y, Cb, Cr, pixels) in one arena, byte offsets; us per call")
ref = (0, szY, szY + szC, szY + 2 * szC)
for rep in range(2):
    print(f"contiguous                       : {run(*ref):6.1f}")
for d in [256, 1 * K, 4 * K, 16 * K, 64 * K, 256 * K, 1 * MiB, 2 * MiB, 8 * MiB, 32 * MiB, 64 * MiB, 96 * MiB, 128 * MiB, 256 * MiB, 512 * MiB]:
    print(f"pixels +{d:>10d}               : {run(ref[0], ref[1], ref[2], ref[3] + d):6.1f}")
for d in [4 * K, 64 * K, 1 * MiB, 16 * MiB, 64 * MiB, 512 * MiB]:
    print(f"Cb, Cr +{d:>10d} (pixels +1 GiB): {run(0, szY + d, szY + szC + 2 * d, 1024 * MiB + szY + 2 * szC):6.1f}")
for d in [0, 4 * K, 64 * K, 1 * MiB, 16 * MiB]:
    print(f"Cr +{d:>10d} only             : {run(0, szY, szY + szC + d, 2048 * MiB):6.1f}")
for rep in range(2):
    print(f"contiguous                       : {run(*ref):6.1f}")
