"""BASELINE.json configs[0]: a single 512x512 baseline ycc8 4:2:0 image through the CPU path --
plumbing only, no GPU.  pixels -> oracle encode -> host entropy encoder (file bytes) -> host
entropy decoder -> oracle decode -> pixels; the independent Python reader must see the same file."""
import ctypes as C

import numpy as np

from jpeg_amd import _lib, compression_quanta
from jpeg_amd.api import _scan_array, _metadata_array
from oracle import jpeg_reader, oracle as O


def _picture(w, h):
    yy, xx = np.mgrid[0:h, 0:w]
    rng = np.random.default_rng(20240807)
    base = 128 + 70 * np.sin(xx / 37.0) * np.cos(yy / 23.0)
    rgb = base[..., None] + np.array([10, -5, 20]) + rng.integers(-8, 9, (h, w, 3))
    return np.clip(rgb, 0, 255).astype(np.uint8).reshape(-1, 3)


def test_config1_512x512_baseline_420_cpu_round_trip(tmp_path):
    w = h = 512
    factors = [(2, 2), (1, 1), (1, 1)]
    q = [compression_quanta("luminance", 1.0), compression_quanta("chrominance", 1.0)]
    rgb = _picture(w, h)
    coef = O.encode(rgb, (w, h), factors, [q[0], q[1], q[1]])
    assert [c.shape for c in coef] == [(64, 64, 64), (32, 32, 64), (32, 32, 64)]      # 6144 blocks, SURVEY 8

    lib = _lib.lib()
    info = _lib.FrameInfo()
    info.width, info.height, info.precision, info.ncomponents, info.process = w, h, 8, 3, 0
    for c, (fx, fy) in enumerate(factors):
        info.id[c], info.factor_x[c], info.factor_y[c] = c + 1, fx, fy
        info.units_y[c], info.units_x[c] = coef[c].shape[:2]
    tables = np.stack(q).astype(np.uint16)
    sarr = _scan_array([[(0, 0, 0), (1, 1, 1), (2, 1, 1)]])
    marr, nmeta, _keep = _metadata_array([("jfif", (2, 1, 72, 72))])
    out, n = np.empty(1 << 20, np.uint8), C.c_size_t()
    planes = [np.ascontiguousarray(c) for c in coef]
    st = lib.jpeg_amd_jpeg_encode_spectral(C.byref(info), (C.c_int32 * 3)(0, 1, 1), _lib.ptr_array([p.ctypes.data for p in planes]),
                                           tables.ctypes.data, (C.c_int32 * 2)(0, 1), 2, sarr, 1, marr, nmeta,
                                           out.ctypes.data, out.size, C.byref(n))
    assert st == 0
    path = tmp_path / "c1.jpg"
    path.write_bytes(out[:n.value].tobytes())

    # host entropy decoder of the library
    data = out[:n.value].copy()
    info2 = _lib.FrameInfo()
    back = [np.zeros_like(p) for p in planes]
    q2 = np.zeros((4, 64), np.uint16)
    assert lib.jpeg_amd_jpeg_decode_spectral(data.ctypes.data, data.size, _lib.ptr_array([p.ctypes.data for p in back]),
                                             q2.ctypes.data, C.byref(info2)) == 0
    assert (info2.width, info2.height, info2.process, info2.nscans) == (w, h, 0, 1)
    for a, b in zip(planes, back):
        assert (a == b).all()
    # the independent (test-only) Python reader agrees
    img = jpeg_reader.read_jpeg(str(path))
    for a, b in zip(planes, img.planes):
        assert (a == np.asarray(b)).all()
    # and the picture survives: decode on the CPU path
    _, rect = O.decode(back, [q2[0], q2[1], q2[2]], factors, (w, h))
    got = O.unpack_rgb8(rect, 3)
    err = got.astype(np.float64) - rgb.astype(np.float64)
    assert 10 * np.log10(255.0 ** 2 / np.mean(err ** 2)) > 30.0
