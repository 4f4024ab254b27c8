#!/usr/bin/env python3
"""Restart-interval-parallel host entropy decoding of ONE large file, written by this library with
`restart_interval` (an extension; the reference's writer never emits DRI) -- or by Pillow without a GPU."""
import ctypes as C, io, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from jpeg_amd import _lib
lib = _lib.lib()
try:                      # the GPU part is optional: create the context before anything else
    import jpeg_amd as J
    ctx = J.Context(0)
except Exception as e:    # noqa: BLE001
    ctx = None
    print("no GPU context:", repr(e))
W = H = 8192
yy, xx = np.mgrid[0:H, 0:W]
rng = np.random.default_rng(1)
img = np.clip(128 + 70 * np.sin(xx / 37.0) * np.cos(yy / 23.0) + rng.integers(-10, 11, (H, W)), 0, 255).astype(np.uint8)
rgb = np.stack([img, np.roll(img, 5, 0), np.roll(img, 9, 1)], -1)
if ctx is not None:      # this library's own writer with one restart interval per MCU row (an extension)
    layout = J.Layout("ycc8", {1: ((2, 2), 0), 2: ((1, 1), 1), 3: ((1, 1), 1)})
    quanta = {0: J.compression_quanta("luminance", 0.5), 1: J.compression_quanta("chrominance", 0.5)}
    raw = J.Rectangular.pack(ctx, (W, H), layout, rgb.reshape(-1, 3), J.RGB).compress(
        quanta, [[(0, 0, 0), (1, 1, 1), (2, 1, 1)]], restart_interval=W // 16)
    data = np.frombuffer(raw, np.uint8).copy()
else:                    # no GPU: Pillow writes the file
    from PIL import Image
    buf = io.BytesIO(); Image.fromarray(rgb).save(buf, format="JPEG", quality=85, subsampling=2, restart_marker_rows=1)
    data = np.frombuffer(buf.getvalue(), np.uint8).copy()
info = _lib.FrameInfo(); assert lib.jpeg_amd_jpeg_inspect(data.ctypes.data, data.size, C.byref(info)) == 0
print(f"{W}x{H} 4:2:0 baseline, {data.size/1e6:.1f} MB, restart interval {info.restart_interval} MCUs")
planes = [np.ones((info.units_y[c], info.units_x[c], 64), np.int16) for c in range(3)]; q = np.zeros((4, 64), np.uint16)
ref = None
for t in (1, 2, 4, 8, 16, 32, 64):
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        st = lib.jpeg_amd_jpeg_decode_spectral_mt(data.ctypes.data, data.size, _lib.ptr_array([p.ctypes.data for p in planes]), q.ctypes.data, None, t)
        best = min(best, time.perf_counter() - t0)
    h = hash(b"".join(p.tobytes() for p in planes))
    ref = ref if ref is not None else h
    print(f"  {t:3d} threads: {best*1e3:7.1f} ms  {W*H/best/1e6:8.0f} Mpx/s  {data.size/best/1e6:7.0f} MB/s  same planes: {h == ref and st == 0}")

# the whole call, file bytes in host memory -> RGB bytes in host memory (needs the GPU)
if ctx is not None:
    out = np.empty(W * H * 3, np.uint8)
    for rep in range(3):
        t0 = time.perf_counter()
        st = lib.jpeg_amd_decompress(ctx.handle, data.ctypes.data, data.size, 0, J.RGB.code, out.ctypes.data, out.size, None)
        dt = time.perf_counter() - t0
        assert st == 0, st
    print(f"jpeg_amd_decompress end to end: {dt*1e3:.1f} ms = {W*H/dt/1e6:.0f} Mpx/s (entropy decode on all cores, H2D 201 MB, kernels, D2H 201 MB)")
