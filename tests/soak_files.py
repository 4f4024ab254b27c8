#!/usr/bin/env python3
"""Soak of the batch FILE paths (SURVEY 8f-1 ... f-3): random geometry, layout, restart interval, batch size and thread count
through jpeg_amd_compress_batch / _device (sparse download + jpeg_amd_jpeg_encode_sparse) and jpeg_amd_decompress_batch / _device
(jpeg_amd_jpeg_decode_sparse + k_expand_sparse), against the single-picture entry points and the staged mirror.
    python tests/soak_files.py <seed> <cases>   (not collected by pytest)"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import jpeg_amd as J
from jpeg_amd import _lib
from jpeg_amd.api import _scan_array, _metadata_array
ctx = J.Context(0); lib = _lib.lib()
rng = np.random.default_rng(int(sys.argv[1]))
N = int(sys.argv[2])
LAYOUTS = {"grey": [(1, 1)], "444": [(1, 1)] * 3, "420": [(2, 2), (1, 1), (1, 1)], "422": [(2, 1), (1, 1), (1, 1)], "440": [(1, 2), (1, 1), (1, 1)]}
bad = 0
for it in range(N):
    name = list(LAYOUTS)[int(rng.integers(len(LAYOUTS)))]
    factors = LAYOUTS[name]
    nc = len(factors)
    w, h = int(rng.integers(1, 500)), int(rng.integers(1, 300))
    n = int(rng.choice([1, 2, 5, 33, 70, 150]))
    threads = int(rng.choice([1, 2, 3, 8, 16, 40]))
    ri = int(rng.choice([0, 0, 1, 7, 64]))
    level = float(rng.choice([0.25, 1.0, 4.0]))
    dense = rng.integers(6) == 0                      # now and then pictures too dense for the sparse arenas
    yy, xx = np.mgrid[0:h, 0:w]
    px = np.zeros((n, h, w, 3), np.uint8)
    for i in range(n):
        base = 128 + 80 * np.sin(xx / (7.0 + i)) * np.cos(yy / (5.0 + 2 * i))
        noise = rng.integers(-90, 91, (h, w, 3)) if (dense and i % 2 == 0) else rng.integers(-6, 7, (h, w, 3))
        px[i] = np.clip(base[..., None] + noise + np.array([0, 20, -20]), 0, 255)
    info = _lib.FrameInfo()
    info.width, info.height, info.precision, info.ncomponents, info.process = w, h, 8, nc, 0
    info.restart_interval = ri
    for c, (fx, fy) in enumerate(factors):
        info.id[c], info.factor_x[c], info.factor_y[c] = c + 1, fx, fy
    if dense:
        tables = np.ones((2, 64), np.uint16)
    else:
        tables = np.stack([J.compression_quanta("luminance", level), J.compression_quanta("chrominance", level)]).astype(np.uint16)
    qkey = (C.c_int32 * nc)(*([0] + [1] * (nc - 1)))
    tk = (C.c_int32 * 2)(0, 1)
    scans = [[(0, 0, 0)]] if nc == 1 else ([[(0, 0, 0)], [(1, 1, 1), (2, 1, 1)]] if rng.integers(2) else [[(0, 0, 0), (1, 1, 1), (2, 1, 1)]])
    sarr = _scan_array(scans)
    marr, nmeta, _keep = _metadata_array([("jfif", (2, 2, 1, 1))])
    cap = 12 * (w + 16) * (h + 16) + 8192      # (noise under all-ones tables is larger than its pixels)
    def compress(fn, src, stride):
        out = np.zeros((n, cap), np.uint8); sizes = (C.c_size_t * n)()
        f = _lib.FrameInfo(); C.memmove(C.byref(f), C.byref(info), C.sizeof(f))
        st = fn(ctx.handle, C.byref(f), src, stride, n, J.RGB.code, qkey, tables.ctypes.data, tk, 2, sarr, len(scans), marr, nmeta, threads,
                out.ctypes.data, cap, sizes)
        assert st == 0, (st, name, w, h, n)
        return [out[i, :sizes[i]].tobytes() for i in range(n)]
    files = compress(lib.jpeg_amd_compress_batch, px.ctypes.data, 0)
    d_px = torch.from_numpy(px.reshape(-1)).to(ctx.torch_device)
    files_dev = compress(lib.jpeg_amd_compress_batch_device, d_px.data_ptr(), 0)
    ok = files == files_dev
    # a picture through the single-picture entry point (a batch of one takes the same code with other chunk sizes)
    i0 = int(rng.integers(n))
    one = np.zeros(cap, np.uint8); nb = C.c_size_t()
    f1 = _lib.FrameInfo(); C.memmove(C.byref(f1), C.byref(info), C.sizeof(f1))
    st = lib.jpeg_amd_compress_batch(ctx.handle, C.byref(f1), px[i0].ctypes.data, 0, 1, J.RGB.code, qkey, tables.ctypes.data, tk, 2, sarr, len(scans),
                                     marr, nmeta, 1, one.ctypes.data, cap, C.byref(nb))
    ok = ok and st == 0 and one[:nb.value].tobytes() == files[i0]
    # ... and through the staged mirror (planes on the host, the writer on planes), when no restart interval is asked for
    if ri == 0:
        layout = J.Layout("y8" if nc == 1 else "ycc8", {c + 1: (tuple(f), min(c, 1)) for c, f in enumerate(factors)})
        quanta = {0: tables[0], 1: tables[1]}
        want = J.Rectangular.pack(ctx, (w, h), layout, px[i0].reshape(-1, 3), J.RGB).decomposed().fdct(quanta).compress(scans, metadata=[("jfif", (2, 2, 1, 1))])
        ok = ok and want == files[i0]
    # decode: batch to host, batch to device, single
    bufs = [np.frombuffer(f, np.uint8).copy() for f in files]
    ptrs = (C.c_void_p * n)(*[b.ctypes.data for b in bufs]); sizes = (C.c_size_t * n)(*[b.size for b in bufs])
    stride = w * h * 3 + int(rng.integers(0, 3)) * 8
    out_h = np.zeros(n * stride, np.uint8)
    st = lib.jpeg_amd_decompress_batch(ctx.handle, ptrs, sizes, n, threads, 0, J.RGB.code, out_h.ctypes.data, stride, None)
    ok = ok and st == 0
    out_d = torch.zeros(n * stride, dtype=torch.uint8, device=ctx.torch_device)
    st = lib.jpeg_amd_decompress_batch_device(ctx.handle, ptrs, sizes, n, threads, 0, J.RGB.code, out_d.data_ptr(), stride, None)
    ok = ok and st == 0 and (out_d.cpu().numpy() == out_h).all()
    single = np.zeros(w * h * 3, np.uint8)
    st = lib.jpeg_amd_decompress(ctx.handle, bufs[i0].ctypes.data, bufs[i0].size, 0, J.RGB.code, single.ctypes.data, single.size, None)
    ok = ok and st == 0 and (single == out_h[i0 * stride:i0 * stride + w * h * 3]).all()
    if not ok:
        bad += 1
        print("MISMATCH", name, w, h, n, threads, ri, level, dense, len(scans), flush=True)
print("file soak done", N, "cases, mismatches:", bad)
sys.exit(1 if bad else 0)
