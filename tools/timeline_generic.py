#!/usr/bin/env python3
"""Start, end and placement (XCC, SE, CU) of every sampled workgroup of k_generic_fused on one 8192 x 8192 4:2:0 12-bit image: who is resident
when, who leaves last and in which phase it lost its time (needs a -DJA_GEN_PHASE build):
    tools/build_exp.sh gphase -DJA_GEN_PHASE; JPEG_AMD_LIBRARY=tools/exp/libjpeg_amd_gphase.so python tools/timeline_generic.py"""
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
for _ in range(4): step()
torch.cuda.synchronize(); ctx.timer_begin(); step(); ms = ctx.timer_end()
fn = lib.jpeg_amd_debug_gen_phase; fn.restype = C.c_int; fn.argtypes = [C.c_void_p, C.c_size_t]
buf = np.zeros((4096, 16), np.uint64); assert fn(buf.ctypes.data, buf.size) == 0
buf = buf[buf[:, 14] > 0]
start = buf[:, 12].astype(np.float64) * 0.01; life = buf[:, 15].astype(np.float64) * 0.01
start -= start.min(); end = start + life
xcc = (buf[:, 13] >> np.uint64(32)).astype(int); hw = (buf[:, 13] & np.uint64(0xffffffff)).astype(np.int64)
cu = (hw >> 8) & 15; sh = (hw >> 12) & 1; se = (hw >> 13) & 7; simd = (hw >> 4) & 3
print(f"step {ms*1e3:.1f} us; {len(buf)} workgroups sampled; start: {np.sum(start < 10)} within 10 us, {np.sum(start >= 10)} later (median of the late ones {np.median(start[start >= 10]) if np.any(start >= 10) else 0:.1f} us)")
print(f"life: mean {life.mean():.1f} median {np.median(life):.1f} min {life.min():.1f} max {life.max():.1f} us; end: median {np.median(end):.1f}, 90% {np.percentile(end, 90):.1f}, max {end.max():.1f} us")
for x in range(8):
    m = xcc == x
    if m.any(): print(f"  xcc {x}: {m.sum():4d} wgs, life mean {life[m].mean():6.1f}  end max {end[m].max():6.1f}  late starters {np.sum(start[m] >= 10)}")
key = xcc * 100000 + se * 1000 + sh * 100 + cu
vals, counts = np.unique(key, return_counts=True)
print("sampled workgroups per CU:", dict(zip(*np.unique(counts, return_counts=True))))
order = np.argsort(-life)[:6]
names = ["geometry", "tables", "barrier 1", "coef wait", "IDCT", "barrier 2", "gather", "staging"]
for i in order:
    same = key == key[i]
    print(f"slow wg: life {life[i]:.1f} us xcc {xcc[i]} se {se[i]} sh {sh[i]} cu {cu[i]}; its CU mates' lives: {np.round(life[same], 1).tolist()}; phases (k cycles): " + ", ".join(f"{n} {buf[i, j] / 1e3:.0f}" for j, n in enumerate(names)))
