#!/usr/bin/env python3
"""Per-phase wall cycles of k_luma_fused's waves (needs a -DJA_PHASE_PROFILE build):
    tools/build_exp.sh prof -DJA_PHASE_PROFILE
    JPEG_AMD_LIBRARY=tools/exp/libjpeg_amd_prof.so python tools/phase_profile.py"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import jpeg_amd as J
from jpeg_amd import _lib, synth

ctx = J.Context(0); dev = ctx.torch_device; lib = _lib.lib()
W = H = 8192
layout = J.Layout("ycc8", {1: J.Component((2, 2), 0), 2: J.Component((1, 1), 1), 3: J.Component((1, 1), 1)})
units = layout.units((W, H)); L = layout.c_layout((W, H), units, [0, 1, 1])
q_np = np.stack([J.compression_quanta("luminance", 1.0), J.compression_quanta("chrominance", 1.0)])
d_q = torch.from_numpy(q_np.view(np.int16)).to(dev)
planes = synth.natural_planes_torch(units, 2, dev, 3)
out = torch.empty((2, W * H * 3), dtype=torch.uint8, device=dev)
strides = _lib.size_array([64 * a * b for a, b in units])
def step(r):
    st = lib.jpeg_amd_decode_batch(ctx.handle, C.byref(L), 1, _lib.ptr_array([p[r].data_ptr() for p in planes]), strides,
                                   d_q.data_ptr(), 0, 2, 0, _lib.COLOR_RGB8, out[r].data_ptr(), W * H * 3)
    assert st == 0
for i in range(4): step(i & 1)
torch.cuda.synchronize()
ctx.timer_begin(); step(0); ms = ctx.timer_end()
fn = lib.jpeg_amd_debug_phase_cycles; fn.restype = C.c_int; fn.argtypes = [C.c_void_p, C.c_size_t]
buf = np.zeros((3072, 8), np.uint64)
assert fn(buf.ctypes.data, buf.size) == 0
names = ["wait coef DMA", "LDS coef read + chroma DMA issue", "IDCT", "chroma wait + next DMA issue", "hrow prologue + geometry", "8 pixel rows (colour, stage, store)"]
tot = buf[:, :6].sum(axis=1).astype(np.float64)
print(f"step {ms*1e3:.1f} us; per-wave total cycles mean {tot.mean():.0f} min {tot.min():.0f} max {tot.max():.0f}")
for i, n in enumerate(names):
    c = buf[:, i].astype(np.float64)
    print(f"  {n:40s} {c.mean():10.0f} cycles/wave  {100*c.mean()/tot.mean():5.1f} %   per strip {c.mean()/(16384/3072):8.0f}")
life, ticks = buf[:, 6].astype(np.float64), buf[:, 7].astype(np.float64)
print(f"wave life: mean {life.mean():.0f} max {life.max():.0f} shader cycles = mean {ticks.mean() / 100:.1f} max {ticks.max() / 100:.1f} us of the 100 MHz counter"
      f" -> effective shader clock {100.0 * (life / ticks).mean():.0f} MHz while k_luma_fused runs")
