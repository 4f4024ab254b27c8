"""oracle.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

ctypes/numpy front end of oracle/libjpeg_oracle.so (the C restatement of the
reference's hot path, see jpeg_oracle.h).  Only tests/, __graft_entry__.smoke()
and bench.py's cpu_baseline leg may import this module.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from concurrent.futures import ThreadPoolExecutor

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

_u16p = C.POINTER(C.c_uint16)
_i16p = C.POINTER(C.c_int16)
_u8p = C.POINTER(C.c_uint8)
_intp = C.POINTER(C.c_int)


def build(force: bool = False, target: str = "libjpeg_oracle.so") -> str:
    """Compile the C restatement with oracle/Makefile (gcc, -ffp-contract=off)."""
    path = os.path.join(_HERE, target)
    src = [os.path.join(_HERE, f) for f in ("jpeg_oracle.c", "jpeg_oracle.h", "Makefile")]
    stale = (not os.path.exists(path)) or any(
        os.path.getmtime(s) > os.path.getmtime(path) for s in src)
    if force or stale:
        subprocess.check_call(["make", "-C", _HERE, "-B" if force else "-s", target],
                              stdout=subprocess.DEVNULL)
    return path


def lib() -> C.CDLL:
    global _LIB
    if _LIB is None:
        L = C.CDLL(build())
        L.orc_zigzag.argtypes = [C.c_int, C.c_int]
        L.orc_zigzag.restype = C.c_int
        L.orc_modulate.argtypes = [_u16p, C.c_float, C.POINTER(C.c_float)]
        L.orc_idct_plane_rows.argtypes = [_i16p, C.c_int, C.c_int, _u16p, C.c_int, _u16p,
                                          C.c_int, C.c_int]
        L.orc_interleave_rows.argtypes = [C.POINTER(_u16p), _intp, _intp, _intp, _intp,
                                          C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                          C.c_int, _u16p, C.c_int, C.c_int]
        for f in ("orc_unpack_rgb8", "orc_unpack_ycc8", "orc_pack_rgb8", "orc_pack_ycc8"):
            getattr(L, f).argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
        L.orc_decompose_plane.argtypes = [_u16p, C.c_int, C.c_int, C.c_int, C.c_int,
                                          C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                          C.c_int, _u16p]
        L.orc_fdct_plane_rows.argtypes = [_u16p, C.c_int, C.c_int, _u16p, C.c_int, _i16p,
                                          C.c_int, C.c_int]
        L.orc_compression_quanta.argtypes = [C.c_int, C.c_double, _u16p]
        for f in ("orc_modulate", "orc_idct_plane_rows", "orc_interleave_rows",
                  "orc_unpack_rgb8", "orc_unpack_ycc8", "orc_pack_rgb8", "orc_pack_ycc8",
                  "orc_decompose_plane", "orc_fdct_plane_rows", "orc_compression_quanta"):
            getattr(L, f).restype = None
        _LIB = L
    return _LIB


def _ptr(a: np.ndarray, typ):
    return a.ctypes.data_as(typ)


def _c(a, dtype):
    a = np.ascontiguousarray(a, dtype=dtype)
    return a


def units(size: int, stride: int) -> int:
    """decode.swift:1364-1369"""
    return size // stride + (1 if size % stride else 0)


def plane_units(size, factor, scale):
    """decode.swift:2606-2616: ceil(size * factor / (8 * scale)) per axis."""
    return (units(size[0] * factor[0], 8 * scale[0]),
            units(size[1] * factor[1], 8 * scale[1]))


def zigzag(k: int, h: int) -> int:
    return lib().orc_zigzag(k, h)


def modulate(q_zz, scale: float) -> np.ndarray:
    q = _c(q_zz, np.uint16)
    out = np.empty(64, np.float32)
    lib().orc_modulate(_ptr(q, _u16p), scale, _ptr(out, C.POINTER(C.c_float)))
    return out.reshape(8, 8)


def _bands(n: int, threads: int):
    threads = max(1, min(threads, n))
    edges = [n * i // threads for i in range(threads + 1)]
    return [(a, b) for a, b in zip(edges[:-1], edges[1:]) if b > a]


def idct_plane(coef, q_zz, precision: int = 8, threads: int = 1) -> np.ndarray:
    """coef int16 [uy, ux, 64] zigzag -> uint16 [8uy, 8ux]."""
    coef = _c(coef, np.int16)
    uy, ux, _ = coef.shape
    q = _c(q_zz, np.uint16)
    out = np.empty((8 * uy, 8 * ux), np.uint16)
    L = lib()

    def run(band):
        L.orc_idct_plane_rows(_ptr(coef, _i16p), ux, uy, _ptr(q, _u16p), precision,
                              _ptr(out, _u16p), band[0], band[1])
    if threads <= 1:
        run((0, uy))
    else:
        with ThreadPoolExecutor(threads) as ex:
            list(ex.map(run, _bands(uy, threads)))
    return out


def interleave(planes, factors, scale, size, cosited: bool = False,
               threads: int = 1) -> np.ndarray:
    """planes: list of uint16 [8uy, 8ux]; -> uint16 [H, W, count]."""
    count = len(planes)
    planes = [_c(p, np.uint16) for p in planes]
    W, H = size
    ux = (C.c_int * count)(*[p.shape[1] // 8 for p in planes])
    uy = (C.c_int * count)(*[p.shape[0] // 8 for p in planes])
    fx = (C.c_int * count)(*[f[0] for f in factors])
    fy = (C.c_int * count)(*[f[1] for f in factors])
    pp = (_u16p * count)(*[_ptr(p, _u16p) for p in planes])
    out = np.empty((H, W, count), np.uint16)
    L = lib()

    def run(band):
        L.orc_interleave_rows(pp, ux, uy, fx, fy, count, scale[0], scale[1], W, H,
                              1 if cosited else 0, _ptr(out, _u16p), band[0], band[1])
    if threads <= 1:
        run((0, H))
    else:
        with ThreadPoolExecutor(threads) as ex:
            list(ex.map(run, _bands(H, threads)))
    return out


def _chunked(fn, src, dst, npx, in_per_px, out_per_px, ncomp, threads):
    sflat, dflat = src.reshape(-1), dst.reshape(-1)

    def run(band):
        a, b = band
        fn(sflat.ctypes.data + a * in_per_px * sflat.itemsize, b - a, ncomp,
           dflat.ctypes.data + a * out_per_px * dflat.itemsize)
    if threads <= 1:
        run((0, npx))
    else:
        with ThreadPoolExecutor(threads) as ex:
            list(ex.map(run, _bands(npx, threads)))


def _px_fn(name):
    return getattr(lib(), name)


def unpack_rgb8(values, ncomp: int, threads: int = 1) -> np.ndarray:
    v = _c(values, np.uint16)
    npx = v.size // ncomp
    out = np.empty((npx, 3), np.uint8)
    _chunked(_px_fn("orc_unpack_rgb8"), v, out, npx, ncomp, 3, ncomp, threads)
    return out


def unpack_ycc8(values, ncomp: int, threads: int = 1) -> np.ndarray:
    v = _c(values, np.uint16)
    npx = v.size // ncomp
    out = np.empty((npx, 3), np.uint8)
    _chunked(_px_fn("orc_unpack_ycc8"), v, out, npx, ncomp, 3, ncomp, threads)
    return out


def pack_rgb8(rgb, ncomp: int, threads: int = 1) -> np.ndarray:
    v = _c(rgb, np.uint8)
    npx = v.size // 3
    out = np.empty((npx, ncomp), np.uint16)
    _chunked(_px_fn("orc_pack_rgb8"), v, out, npx, 3, ncomp, ncomp, threads)
    return out


def pack_ycc8(ycc, ncomp: int, threads: int = 1) -> np.ndarray:
    v = _c(ycc, np.uint8)
    npx = v.size // 3
    out = np.empty((npx, ncomp), np.uint16)
    _chunked(_px_fn("orc_pack_ycc8"), v, out, npx, 3, ncomp, ncomp, threads)
    return out


def decompose(values, size, factors, scale) -> list:
    """values uint16 [H, W, count] -> list of planes uint16 [8uy, 8ux]."""
    W, H = size
    v = _c(values, np.uint16).reshape(H, W, -1)
    count = v.shape[2]
    planes = []
    for p, f in enumerate(factors):
        ux, uy = plane_units(size, f, scale)
        out = np.empty((8 * uy, 8 * ux), np.uint16)
        lib().orc_decompose_plane(_ptr(v, _u16p), W, H, count, p, f[0], f[1],
                                  scale[0], scale[1], ux, uy, _ptr(out, _u16p))
        planes.append(out)
    return planes


def fdct_plane(plane, q_zz, precision: int = 8, threads: int = 1) -> np.ndarray:
    """plane uint16 [8uy, 8ux] -> int16 [uy, ux, 64] zigzag."""
    plane = _c(plane, np.uint16)
    uy, ux = plane.shape[0] // 8, plane.shape[1] // 8
    q = _c(q_zz, np.uint16)
    out = np.empty((uy, ux, 64), np.int16)
    L = lib()

    def run(band):
        L.orc_fdct_plane_rows(_ptr(plane, _u16p), ux, uy, _ptr(q, _u16p), precision,
                              _ptr(out, _i16p), band[0], band[1])
    if threads <= 1:
        run((0, uy))
    else:
        with ThreadPoolExecutor(threads) as ex:
            list(ex.map(run, _bands(uy, threads)))
    return out


def compression_quanta(kind: str, level: float) -> np.ndarray:
    out = np.empty(64, np.uint16)
    lib().orc_compression_quanta(0 if kind == "luminance" else 1, float(level),
                                 _ptr(out, _u16p))
    return out


# ---- whole-path conveniences (mirror the reference's staged calls) ----------

def decode(planes, quanta, factors, size, precision: int = 8, cosited: bool = False,
           threads: int = 1, scale=None):
    """Spectral -> (planar planes, rectangular uint16 [H, W, count]).
    = spectral.idct().interleaved(cosite:)  (decode.swift:4154, 4182)"""
    if scale is None:
        scale = (max(f[0] for f in factors), max(f[1] for f in factors))
    planar = [idct_plane(c, q, precision, threads) for c, q in zip(planes, quanta)]
    rect = interleave(planar, factors, scale, size, cosited, threads)
    return planar, rect


def encode(rgb, size, factors, quanta, precision: int = 8, threads: int = 1, scale=None):
    """[RGB] -> list of coefficient planes.
    = Rectangular.pack(...).decomposed().fdct(quanta:)  (encode.swift:456, 389, 353)"""
    if scale is None:
        scale = (max(f[0] for f in factors), max(f[1] for f in factors))
    ncomp = len(factors)
    rect = pack_rgb8(rgb, ncomp, threads).reshape(size[1], size[0], ncomp)
    planar = decompose(rect, size, factors, scale)
    return [fdct_plane(p, q, precision, threads) for p, q in zip(planar, quanta)]
