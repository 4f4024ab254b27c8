"""Unit checks of the CPU oracle itself (CPU only)."""
import numpy as np
import pytest

from oracle import oracle as O

# The reference's own zigzag known-answer table: tests/unit/tests.swift:38-48
# (rows = vertical frequency h, columns = horizontal frequency k).
ZIGZAG = [
    [0, 1, 5, 6, 14, 15, 27, 28],
    [2, 4, 7, 13, 16, 26, 29, 42],
    [3, 8, 12, 17, 25, 30, 41, 43],
    [9, 11, 18, 24, 31, 40, 44, 53],
    [10, 19, 23, 32, 39, 45, 52, 54],
    [20, 22, 33, 38, 46, 51, 55, 60],
    [21, 34, 37, 47, 50, 56, 59, 61],
    [35, 36, 48, 49, 57, 58, 62, 63],
]


def test_zigzag_matches_reference_table():
    for h in range(8):
        for k in range(8):
            assert O.zigzag(k, h) == ZIGZAG[h][k]


def test_zigzag_is_permutation():
    assert sorted(O.zigzag(k, h) for h in range(8) for k in range(8)) == list(range(64))


def test_modulate_association():
    # decode.swift:4004-4016: (r[k] * r[h]) * (scale * Float(Q)), all binary32
    r = np.array([1, 1.387039845, 1.306562965, 1.175875602,
                  1, 0.785694958, 0.541196100, 0.275899379], np.float32)
    q = (np.arange(64) * 3 + 1).astype(np.uint16)
    for scale in (np.float32(0.125), np.float32(8.0)):
        got = O.modulate(q, float(scale))
        for h in range(8):
            for k in range(8):
                row = np.float32(scale * np.float32(q[ZIGZAG[h][k]]))
                exp = np.float32(np.float32(r[k] * r[h]) * row)
                assert got[h, k] == exp


def test_idct_dc_only_block():
    # a DC-only block is flat: every sample = trunc(clamp(dc * q00 + level))
    coef = np.zeros((1, 1, 64), np.int16)
    coef[0, 0, 0] = 40
    q = np.full(64, 2, np.uint16)
    out = O.idct_plane(coef, q, 8)
    assert out.shape == (8, 8)
    assert (out == out[0, 0]).all()
    assert out[0, 0] == int(np.float32(40 * 2 * 0.125) + np.float32(128.5))


def test_idct_clamps_to_precision():
    coef = np.zeros((1, 2, 64), np.int16)
    coef[0, 0, 0] = 2047
    coef[0, 1, 0] = -2048
    q = np.full(64, 16, np.uint16)
    for p, hi in ((8, 255), (12, 4095)):
        out = O.idct_plane(coef, q, p)
        assert (out[:, :8] == hi).all() and (out[:, 8:] == 0).all()


def test_fdct_idct_roundtrip_small_error():
    rng = np.random.default_rng(1)
    plane = rng.integers(0, 256, (16, 24)).astype(np.uint16)
    q = np.ones(64, np.uint16)
    coef = O.fdct_plane(plane, q, 8)
    back = O.idct_plane(coef, q, 8)
    assert np.abs(back.astype(int) - plane.astype(int)).max() <= 1


def test_colour_roundtrip_grey_axis():
    y = np.arange(256, dtype=np.uint16)
    rgb = O.unpack_rgb8(y, 1)
    assert (rgb == np.arange(256)[:, None]).all()
    ycc = O.unpack_ycc8(y, 1)
    assert (ycc[:, 0] == np.arange(256)).all() and (ycc[:, 1:] == 128).all()


def test_threads_identical():
    rng = np.random.default_rng(2)
    coef = rng.integers(-300, 300, (9, 7, 64)).astype(np.int16)
    q = rng.integers(1, 50, 64).astype(np.uint16)
    a = O.idct_plane(coef, q, 8, threads=1)
    b = O.idct_plane(coef, q, 8, threads=4)
    assert (a == b).all()
    assert (O.fdct_plane(a, q, 8, threads=3) == O.fdct_plane(a, q, 8)).all()


def test_asan_build_runs_clean():
    """Same restatement under -fsanitize=address,undefined (CPU only)."""
    import ctypes as C
    import subprocess
    import sys
    import os
    path = O.build(target="libjpeg_oracle_asan.so")
    code = (
        "import ctypes as C, numpy as np\n"
        f"L = C.CDLL({path!r})\n"
        "rng = np.random.default_rng(0)\n"
        "coef = rng.integers(-2048, 2048, (3, 5, 64)).astype(np.int16)\n"
        "q = rng.integers(1, 255, 64).astype(np.uint16)\n"
        "out = np.empty((24, 40), np.uint16)\n"
        "L.orc_idct_plane(coef.ctypes.data_as(C.c_void_p), 5, 3, q.ctypes.data_as(C.c_void_p), 8, out.ctypes.data_as(C.c_void_p))\n"
        "c2 = np.empty((3, 5, 64), np.int16)\n"
        "L.orc_fdct_plane(out.ctypes.data_as(C.c_void_p), 5, 3, q.ctypes.data_as(C.c_void_p), 8, c2.ctypes.data_as(C.c_void_p))\n"
        "print('ok')\n")
    asan = subprocess.check_output(["gcc", "-print-file-name=libasan.so"]).decode().strip()
    env = dict(os.environ, LD_PRELOAD=asan, ASAN_OPTIONS="detect_leaks=0")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True)
    assert r.returncode == 0 and "ok" in r.stdout, r.stderr[-2000:]
