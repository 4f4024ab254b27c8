#!/usr/bin/env python3
"""How many workgroups of k_encode_fused are resident at once?  (needs a -DJA_ENC_TIMELINE build)
    tools/build_exp.sh enctl -DJA_ENC_TIMELINE; JPEG_AMD_LIBRARY=tools/exp/libjpeg_amd_enctl.so python tools/timeline_encode.py [size]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import jpeg_amd as J
from jpeg_amd import _lib
ctx = J.Context(0); dev = ctx.torch_device; lib = _lib.lib()
W = H = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
layout = J.Layout("ycc8", {1: J.Component((2, 2), 0), 2: J.Component((1, 1), 1), 3: J.Component((1, 1), 1)})
units = layout.units((W, H)); L = layout.c_layout((W, H), units, [0, 1, 1])
q_np = np.stack([J.compression_quanta("luminance", 1.0), J.compression_quanta("chrominance", 1.0)])
d_q = torch.from_numpy(q_np.view(np.int16)).to(dev)
px = torch.randint(0, 256, (W * H * 3,), dtype=torch.uint8, device=dev)
coef = [torch.empty(64 * a * b, dtype=torch.int16, device=dev) for a, b in units]
strides = _lib.size_array([64 * a * b for a, b in units])
def step():
    st = lib.jpeg_amd_encode_batch(ctx.handle, C.byref(L), 1, px.data_ptr(), 0, _lib.COLOR_RGB8, d_q.data_ptr(), 0, 2,
                                   _lib.ptr_array([c.data_ptr() for c in coef]), _lib.size_array([0] * len(units)))
    assert st == 0, st
for _ in range(4): step()
torch.cuda.synchronize(); ctx.timer_begin(); step(); ms = ctx.timer_end()
fn = lib.jpeg_amd_debug_encode_timeline; fn.restype = C.c_int; fn.argtypes = [C.c_void_p, C.c_size_t]
buf = np.zeros((32768, 2), np.uint64); assert fn(buf.ctypes.data, buf.size) == 0
buf = buf[buf[:, 1] > 0]
start = buf[:, 0].astype(np.float64) * 0.01; end = buf[:, 1].astype(np.float64) * 0.01
t0 = start.min(); start -= t0; end -= t0; life = end - start
print(f"{W} x {H}: step {ms * 1e3:.1f} us, {len(buf)} workgroups; life mean {life.mean():.2f} us (min {life.min():.2f}, max {life.max():.2f}); last end {end.max():.1f} us")
print(f"  sum of lives {life.sum() / 1e3:.2f} ms / step = {life.sum() / (ms * 1e3):.0f} workgroups resident on average = {life.sum() / (ms * 1e3) / 256:.2f} per CU")
grid = np.arange(0, end.max(), 1.0)
res = [(np.sum((start <= g) & (end > g))) for g in grid]
print("  resident workgroups every 5 us:", [int(r) for r in res[::5]])
