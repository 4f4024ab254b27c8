"""ctypes binding of include/jpeg_amd.h.  There is NO fallback: if the HIP library is
missing or a call fails, this raises."""
from __future__ import annotations

import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
# JPEG_AMD_LIBRARY points at an alternative build of the same library (tools/ experiments)
LIB_PATH = os.environ.get("JPEG_AMD_LIBRARY") or os.path.join(HERE, "libjpeg_amd.so")
MAX_PLANES = 4

OK, EINVAL, ENOMEM, EHIP, ENODEV, ENOSUP = 0, -1, -2, -3, -4, -5
COLOR_YCC8, COLOR_RGB8 = 0, 1
CTX_OWN_STREAM = 1


class JpegAmdError(RuntimeError):
    def __init__(self, status: int, what: str, hip: int = 0):
        self.status, self.hip = status, hip
        msg = lib().jpeg_amd_strerror(status).decode()
        super().__init__(f"{what}: {msg} (status {status}" + (f", hipError {hip})" if hip else ")"))


class Layout(C.Structure):
    """struct jpeg_amd_layout"""
    _fields_ = [
        ("width", C.c_int32), ("height", C.c_int32), ("precision", C.c_int32),
        ("nplanes", C.c_int32), ("scale_x", C.c_int32), ("scale_y", C.c_int32),
        ("factor_x", C.c_int32 * MAX_PLANES), ("factor_y", C.c_int32 * MAX_PLANES),
        ("units_x", C.c_int32 * MAX_PLANES), ("units_y", C.c_int32 * MAX_PLANES),
        ("qi", C.c_int32 * MAX_PLANES),
    ]


_p = C.c_void_p
_pp = C.POINTER(C.c_void_p)
_szp = C.POINTER(C.c_size_t)
_L = C.POINTER(Layout)

# name -> (restype, argtypes); mirrors include/jpeg_amd.h one to one
SIGNATURES = {
    "jpeg_amd_version": (C.c_int, []),
    "jpeg_amd_strerror": (C.c_char_p, [C.c_int]),
    "jpeg_amd_device_count": (C.c_int, [C.POINTER(C.c_int)]),
    "jpeg_amd_ctx_create": (C.c_int, [C.c_int, _p, C.c_int, _pp]),
    "jpeg_amd_ctx_destroy": (C.c_int, [_p]),
    "jpeg_amd_ctx_synchronize": (C.c_int, [_p]),
    "jpeg_amd_last_hip_error": (C.c_int, [_p]),
    "jpeg_amd_layout_units": (C.c_int, [_L]),
    "jpeg_amd_malloc": (C.c_int, [_p, C.c_size_t, _pp]),
    "jpeg_amd_free": (C.c_int, [_p, _p]),
    "jpeg_amd_memcpy_h2d": (C.c_int, [_p, _p, _p, C.c_size_t]),
    "jpeg_amd_memcpy_d2h": (C.c_int, [_p, _p, _p, C.c_size_t]),
    "jpeg_amd_timer_begin": (C.c_int, [_p]),
    "jpeg_amd_timer_end": (C.c_int, [_p, C.POINTER(C.c_float)]),
    "jpeg_amd_idct_plane": (C.c_int, [_p, _p, C.c_int, C.c_int, _p, C.c_int, _p]),
    "jpeg_amd_spectral_idct": (C.c_int, [_p, _L, _pp, _p, C.c_int, _pp]),
    "jpeg_amd_planar_interleaved": (C.c_int, [_p, _L, _pp, C.c_int, _p]),
    "jpeg_amd_rectangular_unpack": (C.c_int, [_p, _p, C.c_size_t, C.c_int, C.c_int, _p]),
    "jpeg_amd_decode_batch": (C.c_int, [_p, _L, C.c_int, _pp, _szp, _p, C.c_size_t, C.c_int,
                                        C.c_int, C.c_int, _p, C.c_size_t]),
    "jpeg_amd_decode": (C.c_int, [_p, _L, _pp, _p, C.c_int, C.c_int, C.c_int, _p]),
    "jpeg_amd_spectral_rectangular_batch": (C.c_int, [_p, _L, C.c_int, _pp, _szp, _p, C.c_size_t, C.c_int, C.c_int, _p, C.c_size_t]),
    "jpeg_amd_spectral_rectangular": (C.c_int, [_p, _L, _pp, _p, C.c_int, C.c_int, _p]),
    "jpeg_amd_host_spectral_rectangular": (C.c_int, [_p, _L, _pp, _p, C.c_int, C.c_int, _p]),
    "jpeg_amd_rectangular_spectral_batch": (C.c_int, [_p, _L, C.c_int, _p, C.c_size_t, _p, C.c_size_t, C.c_int, _pp, _szp]),
    "jpeg_amd_rectangular_spectral": (C.c_int, [_p, _L, _p, _p, C.c_int, _pp]),
    "jpeg_amd_host_rectangular_spectral": (C.c_int, [_p, _L, _p, _p, C.c_int, _pp]),
    "jpeg_amd_rectangular_pack": (C.c_int, [_p, _p, C.c_size_t, C.c_int, C.c_int, _p]),
    "jpeg_amd_rectangular_decomposed": (C.c_int, [_p, _L, _p, _pp]),
    "jpeg_amd_fdct_plane": (C.c_int, [_p, _p, C.c_int, C.c_int, _p, C.c_int, _p]),
    "jpeg_amd_planar_fdct": (C.c_int, [_p, _L, _pp, _p, C.c_int, _pp]),
    "jpeg_amd_encode_batch": (C.c_int, [_p, _L, C.c_int, _p, C.c_size_t, C.c_int, _p,
                                        C.c_size_t, C.c_int, _pp, _szp]),
    "jpeg_amd_encode": (C.c_int, [_p, _L, _p, C.c_int, _p, C.c_int, _pp]),
    "jpeg_amd_host_spectral_idct": (C.c_int, [_p, _L, _pp, _p, C.c_int, _pp]),
    "jpeg_amd_host_planar_interleaved": (C.c_int, [_p, _L, _pp, C.c_int, _p]),
    "jpeg_amd_host_rectangular_unpack": (C.c_int, [_p, _p, C.c_size_t, C.c_int, C.c_int, _p]),
    "jpeg_amd_host_decode": (C.c_int, [_p, _L, _pp, _p, C.c_int, C.c_int, C.c_int, _p]),
    "jpeg_amd_host_rectangular_pack": (C.c_int, [_p, _p, C.c_size_t, C.c_int, C.c_int, _p]),
    "jpeg_amd_host_rectangular_decomposed": (C.c_int, [_p, _L, _p, _pp]),
    "jpeg_amd_host_planar_fdct": (C.c_int, [_p, _L, _pp, _p, C.c_int, _pp]),
    "jpeg_amd_host_encode": (C.c_int, [_p, _L, _p, C.c_int, _p, C.c_int, _pp]),
    "jpeg_amd_huffman_lookup": (C.c_int, [_p, _p, C.c_int, C.c_uint16, _p, _p]),
    "jpeg_amd_huffman_build": (C.c_int, [_p, _p, _p, _p]),
    "jpeg_amd_jpeg_inspect": (C.c_int, [_p, C.c_size_t, _p]),
    "jpeg_amd_jpeg_decode_spectral": (C.c_int, [_p, C.c_size_t, _pp, _p, _p]),
    "jpeg_amd_jpeg_decode_spectral_mt": (C.c_int, [_p, C.c_size_t, _pp, _p, _p, C.c_int]),
    "jpeg_amd_jpeg_decode_spectral_partial": (C.c_int, [_p, C.c_size_t, _pp, _p, _p, C.c_int, C.c_int]),
    "jpeg_amd_spectral_expand_batch": (C.c_int, [_p, _L, C.c_int, _p, C.c_size_t, _p, C.c_size_t, _p, _pp, _szp]),
    "jpeg_amd_jpeg_decode_sparse": (C.c_int, [_p, C.c_size_t, _p, C.c_size_t, _p, C.c_size_t, _p, _p, _p]),
    "jpeg_amd_stream_create": (C.c_void_p, []),
    "jpeg_amd_stream_destroy": (None, [_p]),
    "jpeg_amd_stream_push": (C.c_int, [_p, _p, C.c_size_t, _p, _p]),
    "jpeg_amd_stream_info": (C.c_int, [_p, _p]),
    "jpeg_amd_stream_snapshot": (C.c_int, [_p, _pp, _p]),
    "jpeg_amd_decompress": (C.c_int, [_p, _p, C.c_size_t, C.c_int, C.c_int, _p, C.c_size_t, _p]),
    "jpeg_amd_decompress_rectangular": (C.c_int, [_p, _p, C.c_size_t, C.c_int, C.c_int, C.c_int, _p, C.c_size_t, _p]),
    "jpeg_amd_decompress_batch": (C.c_int, [_p, _pp, _p, C.c_int, C.c_int, C.c_int, C.c_int, _p, C.c_size_t, _p]),
    "jpeg_amd_decompress_batch_device": (C.c_int, [_p, _pp, _p, C.c_int, C.c_int, C.c_int, C.c_int, _p, C.c_size_t, _p]),
    "jpeg_amd_jpeg_encode_spectral": (C.c_int, [_p, _p, _pp, _p, _p, C.c_int, _p, C.c_int, _p, C.c_int, _p, C.c_size_t, _p]),
    "jpeg_amd_jpeg_encode_sparse": (C.c_int, [_p, _p, _p, _p, C.c_size_t, _p, _p, C.c_int, _p, C.c_int, _p, C.c_int, _p, C.c_size_t, _p]),
    "jpeg_amd_compress": (C.c_int, [_p, _p, _p, C.c_int, _p, _p, _p, C.c_int, _p, C.c_int, _p, C.c_int, _p, C.c_size_t, _p]),
    "jpeg_amd_compress_rectangular": (C.c_int, [_p, _p, _p, _p, _p, _p, C.c_int, _p, C.c_int, _p, C.c_int, _p, C.c_size_t, _p]),
    "jpeg_amd_compress_batch": (C.c_int, [_p, _p, _p, C.c_size_t, C.c_int, C.c_int, _p, _p, _p, C.c_int, _p, C.c_int, _p, C.c_int,
                                          C.c_int, _p, C.c_size_t, _p]),
    "jpeg_amd_compress_batch_device": (C.c_int, [_p, _p, _p, C.c_size_t, C.c_int, C.c_int, _p, _p, _p, C.c_int, _p, C.c_int, _p, C.c_int,
                                                 C.c_int, _p, C.c_size_t, _p]),
}


class FrameInfo(C.Structure):
    """struct jpeg_amd_frame_info"""
    _fields_ = [
        ("width", C.c_int32), ("height", C.c_int32), ("precision", C.c_int32), ("ncomponents", C.c_int32),
        ("process", C.c_int32), ("scale_x", C.c_int32), ("scale_y", C.c_int32),
        ("id", C.c_int32 * MAX_PLANES), ("factor_x", C.c_int32 * MAX_PLANES), ("factor_y", C.c_int32 * MAX_PLANES),
        ("units_x", C.c_int32 * MAX_PLANES), ("units_y", C.c_int32 * MAX_PLANES),
        ("nscans", C.c_int32), ("restart_interval", C.c_int32),
    ]


class Scan(C.Structure):
    """struct jpeg_amd_scan"""
    _fields_ = [("ncomponents", C.c_int32), ("component", C.c_int32 * MAX_PLANES),
                ("dc", C.c_int32 * MAX_PLANES), ("ac", C.c_int32 * MAX_PLANES),
                ("band_lo", C.c_int32), ("band_hi", C.c_int32), ("bit", C.c_int32), ("refine", C.c_int32)]


class Jfif(C.Structure):
    """struct jpeg_amd_jfif"""
    _fields_ = [("version_minor", C.c_int32), ("unit", C.c_int32), ("density_x", C.c_int32), ("density_y", C.c_int32)]


class Metadata(C.Structure):
    """struct jpeg_amd_metadata"""
    _fields_ = [("kind", C.c_int32), ("app", C.c_int32), ("jfif", Jfif), ("data", C.c_void_p), ("size", C.c_size_t)]


_LIB = None


def lib() -> C.CDLL:
    """Load libjpeg_amd.so (built by jpeg_amd.build / __graft_entry__.build())."""
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing: run `python -m jpeg_amd.build` (hipcc, gfx950). "
                "jpeg_amd has no CPU fallback.")
        # PyTorch-ROCm ships its own HIP runtime; when both live in one process it has to be the
        # one loaded FIRST, otherwise the context created later sees no device (ENODEV).  The
        # host-only entry points (entropy coder) do not need torch at all: a missing torch is fine.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)  # AttributeError if the header and the library diverge
            fn.restype, fn.argtypes = res, args
        _LIB = L
    return _LIB


def check(status: int, what: str, ctx=None) -> None:
    if status != OK:
        hip = lib().jpeg_amd_last_hip_error(ctx) if ctx else 0
        raise JpegAmdError(status, what, hip)


def ptr_array(ptrs):
    arr = (C.c_void_p * MAX_PLANES)()
    for i, p in enumerate(ptrs):
        arr[i] = p
    return arr


def size_array(vals):
    arr = (C.c_size_t * MAX_PLANES)()
    for i, v in enumerate(vals):
        arr[i] = v
    return arr
