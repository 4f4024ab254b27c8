#!/usr/bin/env python3
"""End-to-end rate of the file path: JPEG bytes in host memory -> RGB bytes in host memory
(host Huffman decode on T threads, PCIe both ways, fused GPU decode).  The files are 1080p
4:2:0 baseline JPEGs written by this library's own encoder from a synthetic frame.
    python tools/bench_files.py [--n 256] [--threads 1 8 64]"""
import argparse, ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import jpeg_amd as J
from jpeg_amd import _lib
ap = argparse.ArgumentParser(); ap.add_argument("--n", type=int, default=256); ap.add_argument("--threads", type=int, nargs="*", default=[1, 8, 16, 32]); ap.add_argument("--pinned", action="store_true", help="the caller's pixel buffers are page-locked: no staging copy")
args = ap.parse_args()
ctx = J.Context(0); lib = _lib.lib()
def timed(call, reps=6):
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); st = call(); ts.append(time.perf_counter() - t0)
        assert st == 0, st
    ts = sorted(ts[1:])          # (the first call also sizes the context's staging buffers and starts its threads)
    return ts[0], ts[len(ts) // 2]
try:
    quota = open('/sys/fs/cgroup/cpu.max').read().split()
    print('CPU bandwidth of this control group (cpu.max):', ' '.join(quota), '=> ' + ('no limit' if quota[0] == 'max' else f'{int(quota[0]) / int(quota[1]):.0f} CPUs') + f'; {os.cpu_count()} hardware threads')
except OSError:
    pass
W, H = 1920, 1080
yy, xx = np.mgrid[0:H, 0:W]
rng = np.random.default_rng(5)
layout = J.Layout("ycc8", {1: ((2, 2), 0), 2: ((1, 1), 1), 3: ((1, 1), 1)})
quanta = {0: J.compression_quanta("luminance", 1.0), 1: J.compression_quanta("chrominance", 1.0)}
files = []
for i in range(8):   # 8 distinct frames, cycled
    base = 128 + 70 * np.sin(xx / (40.0 + 7 * i)) * np.cos(yy / (29.0 + 3 * i))
    rgb = np.clip(base[..., None] + rng.integers(-12, 13, (H, W, 3)) + np.array([0, 10, -10]), 0, 255).astype(np.uint8).reshape(-1, 3)
    data = J.Rectangular.pack(ctx, (W, H), layout, rgb, J.RGB).compress(quanta, [[(0, 0, 0)], [(1, 1, 1), (2, 1, 1)]],
                                                                          metadata=[("jfif", (2, 2, 1, 1))])
    files.append(np.frombuffer(data, np.uint8).copy())
n = args.n
batch = [files[i % 8] for i in range(n)]
ptrs = (C.c_void_p * n)(*[f.ctypes.data for f in batch]); sizes = (C.c_size_t * n)(*[f.size for f in batch])
out_holder = torch.zeros((n, W * H * 3), dtype=torch.uint8, pin_memory=args.pinned); out = out_holder.numpy()
mb = sum(f.size for f in batch) / 1e6
print(f"{n} files of {W}x{H}, {mb/n*1e3:.0f} KB each" + (", the caller's pixel buffers page-locked" if args.pinned else ""))
def line(what, t, best, med, count, with_mb=True):
    print(f"  {what}{t:3d} host threads: best {best*1e3:7.1f} ms = {count/best:8.0f} images/s, median {med*1e3:7.1f} ms = {count/med:8.0f} images/s ({count*W*H/med/1e6:7.0f} Mpx/s" + (f", {mb/med:6.0f} MB/s of JPEG)" if with_mb else ")"))
for t in args.threads:
    best, med = timed(lambda: lib.jpeg_amd_decompress_batch(ctx.handle, ptrs, sizes, n, t, 0, J.RGB.code, out.ctypes.data, 0, None))
    line("", t, best, med, n)
# the same with the pixels left on the device (sparse coefficients up, nothing down)
d_out = torch.zeros((n, W * H * 3), dtype=torch.uint8, device=ctx.torch_device)
for t in args.threads:
    best, med = timed(lambda: lib.jpeg_amd_decompress_batch_device(ctx.handle, ptrs, sizes, n, t, 0, J.RGB.code, d_out.data_ptr(), 0, None))
    line("to device memory ", t, best, med, n)
assert (d_out.cpu().numpy() == out).all()
# host entropy decode alone, one thread
info = _lib.FrameInfo(); f = batch[0]
planes = [np.zeros((a[1], a[0], 64), np.int16) for a in layout.units((W, H))]; q = np.zeros((4, 64), np.uint16)
t0 = time.perf_counter()
for _ in range(20):
    lib.jpeg_amd_jpeg_decode_spectral(f.ctypes.data, f.size, _lib.ptr_array([p.ctypes.data for p in planes]), q.ctypes.data, None)
dt = (time.perf_counter() - t0) / 20
print(f"host entropy decode alone: {dt*1e3:.2f} ms per file on one thread = {W*H/dt/1e6:.0f} Mpx/s, {f.size/dt/1e6:.0f} MB/s")

# the other direction: RGB bytes in host memory -> baseline JPEG bytes in host memory
from jpeg_amd.api import _scan_array, _metadata_array
n = min(args.n, 256)
px_holder = torch.zeros((n, W * H * 3), dtype=torch.uint8, pin_memory=args.pinned); px = px_holder.numpy(); px[:] = rgb.reshape(1, -1)
info = _lib.FrameInfo()
info.width, info.height, info.precision, info.ncomponents, info.process = W, H, 8, 3, 0
for c, (fx, fy) in enumerate([(2, 2), (1, 1), (1, 1)]):
    info.id[c], info.factor_x[c], info.factor_y[c] = c + 1, fx, fy
tables = np.stack([quanta[0], quanta[1]]).astype(np.uint16)
qkey, tk = (C.c_int32 * 3)(0, 1, 1), (C.c_int32 * 2)(0, 1)
sarr = _scan_array([[(0, 0, 0)], [(1, 1, 1), (2, 1, 1)]]); marr, nmeta, _k = _metadata_array([("jfif", (2, 2, 1, 1))])
cap = 1 << 20
jout = np.zeros((n, cap), np.uint8); jsizes = (C.c_size_t * n)()
for t in args.threads:
    best, med = timed(lambda: lib.jpeg_amd_compress_batch(ctx.handle, C.byref(info), px.ctypes.data, 0, n, J.RGB.code, qkey, tables.ctypes.data, tk, 2,
                                                          sarr, 2, marr, nmeta, t, jout.ctypes.data, cap, jsizes))
    line("compress ", t, best, med, n, False)
d_px = torch.from_numpy(px).to(ctx.torch_device)
for t in args.threads:
    best, med = timed(lambda: lib.jpeg_amd_compress_batch_device(ctx.handle, C.byref(info), d_px.data_ptr(), 0, n, J.RGB.code, qkey, tables.ctypes.data, tk, 2,
                                                                 sarr, 2, marr, nmeta, t, jout.ctypes.data, cap, jsizes))
    line("compress from device memory ", t, best, med, n, False)
