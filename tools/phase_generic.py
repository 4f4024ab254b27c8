#!/usr/bin/env python3
"""Per-phase wall cycles of k_generic_fused's waves, one 8192 x 8192 4:2:0 12-bit image (needs a -DJA_GEN_PHASE build):
    tools/build_exp.sh gphase -DJA_GEN_PHASE
    JPEG_AMD_LIBRARY=tools/exp/libjpeg_amd_gphase.so python tools/phase_generic.py"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import jpeg_amd as J
from jpeg_amd import _lib, synth

ctx = J.Context(0); dev = ctx.torch_device; lib = _lib.lib()
W = H = 8192
layout = J.Layout(("custom", 12, 3), {1: J.Component((2, 2), 0), 2: J.Component((1, 1), 1), 3: J.Component((1, 1), 1)})
units = layout.units((W, H)); L = layout.c_layout((W, H), units, [0, 1, 1])
q_np = np.stack([J.compression_quanta("luminance", 1.0), J.compression_quanta("chrominance", 1.0)]).astype(np.uint16)
planes = synth.natural_planes_torch(units, 1, dev, 3)
coef = [(p[0].to(torch.int32) * 16).clamp(-32768, 32767).to(torch.int16) for p in planes]
rect = torch.empty(W * H * 3, dtype=torch.int16, device=dev)
cp = _lib.ptr_array([t.data_ptr() for t in coef])
def step():
    assert lib.jpeg_amd_spectral_rectangular(ctx.handle, C.byref(L), cp, q_np.ctypes.data_as(C.c_void_p), 2, 0, rect.data_ptr()) == 0
back = [torch.empty(64 * a * b, dtype=torch.int16, device=dev) for a, b in units]
bp = _lib.ptr_array([t.data_ptr() for t in back])
def step_encode():
    assert lib.jpeg_amd_rectangular_spectral(ctx.handle, C.byref(L), rect.data_ptr(), q_np.ctypes.data_as(C.c_void_p), 2, bp) == 0
encode = len(sys.argv) > 1 and sys.argv[1] == "encode"
if encode:
    step(); step, names_override = step_encode, True
for _ in range(4): step()
torch.cuda.synchronize()
ctx.timer_begin(); step(); ms = ctx.timer_end()
fn = lib.jpeg_amd_debug_gen_phase; fn.restype = C.c_int; fn.argtypes = [C.c_void_p, C.c_size_t]
buf = np.zeros((4096, 16), np.uint64)
assert fn(buf.ctypes.data, buf.size) == 0
buf = buf[buf[:, 14] > 0]
names = ["geometry + coefficient loads issued", "tables", "barrier 1", "wait for the coefficients", "IDCT + tile write", "barrier 2",
         "pixel passes: gather + filter", "pixel passes: staging + stores"]
if encode:
    names = ["tables + parameters", "A1: rectangular tile -> LDS (incl. the load latency)", "barrier 1", "A2: decomposed() into the plane tiles", "barrier 2",
             "B: FDCT + quantiser + scatter", "barrier 3", "C: blocks out"]
tot = buf[:, 14].astype(np.float64)
real = buf[:, 15].astype(np.float64) * 0.01   # us: ticks of the constant 100 MHz counter
print(f"step {ms * 1e3:.1f} us, {len(buf)} waves sampled (every 8th / 16th); life mean {tot.mean():.0f} cycles, min {tot.min():.0f}, max {tot.max():.0f}; "
      f"in real time {real.mean():.2f} us (the cycle counter runs at {tot.mean() / real.mean() / 1e3:.2f} GHz)")
if encode or real.mean() * 4 < ms * 1e3:   # (a workgroup per tile; the decode's walk keeps its workgroups for the whole call)
    ntiles = (W // 128) * (H // (32 if encode else 64))
    print(f"  {ntiles} workgroups x {real.mean():.2f} us = {ntiles * real.mean() / 1e3:.1f} ms of workgroup life in a step of {ms * 1e3:.1f} us: "
          f"{ntiles * real.mean() / (ms * 1e3):.0f} workgroups resident on average ({ntiles * real.mean() / (ms * 1e3) / 256:.2f} per CU)")
for i, n in enumerate(names):
    c = buf[:, i].astype(np.float64)
    print(f"  {n:40s} {c.mean():9.0f} cycles  {100 * c.mean() / tot.mean():5.1f} %   (min {c.min():.0f}, median {np.median(c):.0f}, max {c.max():.0f})")
