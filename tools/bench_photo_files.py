#!/usr/bin/env python3
"""The file path on one of the reference's own photographs (tests/golden): N copies through jpeg_amd_decompress_batch_device, beside the
file's bits per pixel and its sparse form's entries per block.
    python tools/bench_photo_files.py karlie-kwk-2019.jpg 1024 16     (fixture, files, host threads)"""
import sys, os, time, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import jpeg_amd as J
from jpeg_amd import _lib
import _golden as G
from _sparse import sparse_decode
ctx = J.Context(0); lib = _lib.lib()
name = sys.argv[1]; n = int(sys.argv[2]); threads = int(sys.argv[3])
f = np.fromfile(G.path(G.entry(name)["file"]), np.uint8)
info = _lib.FrameInfo(); lib.jpeg_amd_jpeg_inspect(f.ctypes.data, f.size, C.byref(info))
W, H = info.width, info.height
blocks = sum(info.units_x[c] * info.units_y[c] for c in range(info.ncomponents))
st, desc, ent, q = sparse_decode(lib, f.tobytes(), info)
print(name, W, "x", H, f.size, "bytes,", f"{8*f.size/(W*H):.2f} bits per pixel; sparse status", st, f"entries per block {ent.size/blocks:.1f}" if st == 0 else "")
ptrs = (C.c_void_p * n)(*[f.ctypes.data] * n); sizes = (C.c_size_t * n)(*[f.size] * n)
d_out = torch.zeros((n, W * H * 3), dtype=torch.uint8, device=ctx.torch_device)
ts = []
for rep in range(7):
    t0 = time.perf_counter()
    st = lib.jpeg_amd_decompress_batch_device(ctx.handle, ptrs, sizes, n, threads, 0, J.RGB.code, d_out.data_ptr(), 0, None)
    ts.append(time.perf_counter() - t0); assert st == 0
ts = sorted(ts[1:])
print(f"  {n} files, {threads} threads to device memory: median {ts[len(ts)//2]*1e3:.1f} ms = {n/ts[len(ts)//2]:.0f} images/s = {n*W*H/ts[len(ts)//2]/1e6:.0f} Mpx/s, {n*f.size/ts[len(ts)//2]/1e6:.0f} MB/s of JPEG")
