#!/usr/bin/env python3
"""Host entropy decoder alone, one thread, no GPU ("the true end-to-end bottleneck"): the 1080p 4:2:0 baseline
files of tools/bench_files.py -- a smooth synthetic frame + noise, this library's own writer with optimised Huffman tables,
CompressionLevel 1.0 tables -- rebuilt on the CPU (coefficients from the oracle, which is why this script lives under
tests/), then jpeg_amd_jpeg_decode_spectral timed per file.
    python tests/bench_entropy_cpu.py [reps]   (not collected by pytest)"""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import jpeg_amd as J
from jpeg_amd import _lib, api
from oracle import oracle as O

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
lib = _lib.lib()
W, H = 1920, 1080
yy, xx = np.mgrid[0:H, 0:W]
rng = np.random.default_rng(5)
factors = [(2, 2), (1, 1), (1, 1)]
quanta = [J.compression_quanta("luminance", 1.0), J.compression_quanta("chrominance", 1.0)]


def write_file(rgb):
    planes = O.encode(rgb, (W, H), factors, [quanta[0], quanta[1], quanta[1]], threads=8)
    info = _lib.FrameInfo()
    info.width, info.height, info.precision, info.ncomponents, info.process = W, H, 8, 3, 0
    info.scale_x, info.scale_y, info.restart_interval = 2, 2, 0
    for p, (f, pl) in enumerate(zip(factors, planes)):
        info.id[p] = p + 1
        info.factor_x[p], info.factor_y[p] = f
        info.units_x[p], info.units_y[p] = pl.shape[1], pl.shape[0]
    host = [np.ascontiguousarray(p) for p in planes]
    qkey = (C.c_int32 * 3)(0, 1, 1); tk = (C.c_int32 * 2)(0, 1)
    tables = np.stack(quanta).astype(np.uint16)
    scans = api._scan_array([[(0, 0, 0)], [(1, 1, 1), (2, 1, 1)]])
    marr, nmeta, _keep = api._metadata_array([("jfif", (2, 2, 1, 1))])
    n = C.c_size_t()
    args = [C.byref(info), qkey, _lib.ptr_array([h.ctypes.data for h in host]), tables.ctypes.data, tk, 2, scans, 2, marr, nmeta]
    _lib.check(lib.jpeg_amd_jpeg_encode_spectral(*args, None, 0, C.byref(n)), "size")
    out = np.empty(n.value, np.uint8)
    _lib.check(lib.jpeg_amd_jpeg_encode_spectral(*args, out.ctypes.data, out.size, C.byref(n)), "encode")
    return out, host


files = []
for i in range(4):
    base = 128 + 70 * np.sin(xx / (40.0 + 7 * i)) * np.cos(yy / (29.0 + 3 * i))
    rgb = np.clip(base[..., None] + rng.integers(-12, 13, (H, W, 3)) + np.array([0, 10, -10]), 0, 255).astype(np.uint8).reshape(-1, 3)
    files.append(write_file(rgb))
print(f"{len(files)} files of {W}x{H}, {np.mean([f.size for f, _ in files]) / 1e3:.0f} KB each")
info = _lib.FrameInfo()
best = []
for f, want in files:
    _lib.check(lib.jpeg_amd_jpeg_inspect(f.ctypes.data, f.size, C.byref(info)), "inspect")
    planes = [np.empty((info.units_y[c], info.units_x[c], 64), np.int16) for c in range(3)]
    qout = np.zeros((4, 64), np.uint16)
    ptrs = _lib.ptr_array([p.ctypes.data for p in planes])
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        st = lib.jpeg_amd_jpeg_decode_spectral(f.ctypes.data, f.size, ptrs, qout.ctypes.data, None, 0)
        ts.append(time.perf_counter() - t0)
        assert st == 0, st
    assert all((a == b).all() for a, b in zip(planes, want)), "decoded planes differ from what was encoded"
    best.append(min(ts))
print(f"jpeg_amd_jpeg_decode_spectral, one thread: {np.mean(best) * 1e3:.3f} ms per file (best of {reps}, mean over the files), "
      f"{np.mean([f.size for f, _ in files]) / np.mean(best) / 1e6:.0f} MB/s of JPEG, {W * H / np.mean(best) / 1e6:.0f} Mpx/s")
