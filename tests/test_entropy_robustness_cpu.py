"""Hostile and damaged input to the host entropy decoder: no C++ exception, abort or out-of-range read may come out
of the C ABI (include/jpeg_amd.h: "plain C, no exceptions, no aborts"); damaged entropy-coded data is rendered the way
the reference renders it (an unknown codeword is symbol 0 of length 16, decode.swift:1255-1258).  CPU only."""
import ctypes as C
import time

import numpy as np

import _golden as G
from jpeg_amd import _lib


def sof0(width, height, ncomp=3):
    comps = b"".join(bytes([i + 1, 0x11, 0]) for i in range(ncomp))
    body = bytes([8, height >> 8, height & 255, width >> 8, width & 255, ncomp]) + comps
    return b"\xff\xd8\xff\xc0" + (len(body) + 2).to_bytes(2, "big") + body


def fixture_bytes(name="color-sequential-1.jpg"):
    return open(G.path(G.entry(name)["file"]), "rb").read()


def test_stream_decoder_refuses_a_frame_it_would_need_gigabytes_for():
    # 21 bytes: SOI + SOF0 65535 x 65535, 3 components -- 3 x 8.6 GB of planes for a decoder that owns them
    lib = _lib.lib()
    s = lib.jpeg_amd_stream_create()
    try:
        data = sof0(65535, 65535)
        scans, fin = C.c_int(), C.c_int()
        buf = (C.c_uint8 * len(data)).from_buffer_copy(data)
        st = lib.jpeg_amd_stream_push(s, buf, len(data), C.byref(scans), C.byref(fin))
        assert st in (_lib.ENOMEM, _lib.EINVAL)
        # the error is final: more bytes are not decoded with half-initialised state
        more = (C.c_uint8 * 4)(0xff, 0xd9, 0, 0)
        assert lib.jpeg_amd_stream_push(s, more, 2, C.byref(scans), C.byref(fin)) == st
    finally:
        lib.jpeg_amd_stream_destroy(s)


def test_one_shot_decoder_reports_the_size_without_allocating():
    info = _lib.FrameInfo()
    data = sof0(65535, 65535) + b"\xff\xd9"
    buf = (C.c_uint8 * len(data)).from_buffer_copy(data)
    assert _lib.lib().jpeg_amd_jpeg_inspect(buf, len(data), C.byref(info)) == 0
    assert (info.width, info.height) == (65535, 65535)


def test_a_malformed_table_makes_the_stream_decoder_fail_for_good():
    # a DHT whose counts over-subscribe the code space is rejected as a whole (the table is built into a temporary);
    # a stream decoder that has seen it stays failed -- half-updated tables are never used
    lib = _lib.lib()
    data = fixture_bytes()
    i = data.index(b"\xff\xc4")
    bad_dht = b"\xff\xc4" + (2 + 17 + 3).to_bytes(2, "big") + bytes([0x00, 3] + [0] * 15) + bytes([1, 2, 3])
    hostile = data[:i] + bad_dht + data[i:]
    s = lib.jpeg_amd_stream_create()
    try:
        scans, fin = C.c_int(), C.c_int()
        buf = (C.c_uint8 * len(hostile)).from_buffer_copy(hostile)
        st = lib.jpeg_amd_stream_push(s, buf, len(hostile), C.byref(scans), C.byref(fin))
        assert st == _lib.EINVAL
        assert lib.jpeg_amd_stream_push(s, buf, 0, C.byref(scans), C.byref(fin)) == _lib.EINVAL
    finally:
        lib.jpeg_amd_stream_destroy(s)


def test_damaged_entropy_data_is_rendered_not_refused():
    # overwrite a stretch in the middle of the scan with bytes that contain no 0xFF: codewords become garbage, long
    # runs of 1-bits match no codeword at all.  The reference decodes on (symbol 0, 16 bits); so does this decoder.
    lib = _lib.lib()
    data = bytearray(fixture_bytes())
    sos = data.index(b"\xff\xda")
    start = sos + 2 + int.from_bytes(data[sos + 2:sos + 4], "big")
    rng = np.random.default_rng(7)
    info = _lib.FrameInfo()
    buf0 = (C.c_uint8 * len(data)).from_buffer_copy(bytes(data))
    assert lib.jpeg_amd_jpeg_inspect(buf0, len(data), C.byref(info)) == 0
    clean = None
    for trial in range(6):
        d = bytearray(data)
        if trial:
            lo = start + int(rng.integers(100, len(data) - start - 400))
            d[lo:lo + 200] = bytes(rng.integers(0xf0, 0xff, 200).astype(np.uint8).tolist())
        planes = [np.full((info.units_y[c], info.units_x[c], 64), 7, np.int16) for c in range(info.ncomponents)]
        quanta = np.zeros((4, 64), np.uint16)
        buf = (C.c_uint8 * len(d)).from_buffer_copy(bytes(d))
        st = lib.jpeg_amd_jpeg_decode_spectral(buf, len(d), _lib.ptr_array([p.ctypes.data for p in planes]),
                                               quanta.ctypes.data, None)
        assert st == 0, (trial, st)
        if trial == 0:
            clean = [p.copy() for p in planes]
        else:
            assert any((p != q).any() for p, q in zip(planes, clean))      # the damage is visible ...
            assert (planes[0][0, 0] == clean[0][0, 0]).all()               # ... but not before it


def test_streaming_a_large_scan_in_small_pieces_stays_linear():
    # the search for the end of an incomplete scan resumes where it stopped (it used to restart at the scan's first
    # byte on every push: quadratic in the number of pushes)
    lib = _lib.lib()
    data = fixture_bytes()
    sos = data.index(b"\xff\xda")
    start = sos + 2 + int.from_bytes(data[sos + 2:sos + 4], "big")
    filler = bytes(8 << 20)                      # 8 MiB of zero bits behind the real data: one huge scan
    eoi = data.rindex(b"\xff\xd9")
    big = data[:eoi] + filler + data[eoi:]
    assert start < eoi
    s = lib.jpeg_amd_stream_create()
    try:
        scans, fin = C.c_int(), C.c_int()
        t0 = time.perf_counter()
        step = 4096
        for off in range(0, len(big), step):
            piece = big[off:off + step]
            buf = (C.c_uint8 * len(piece)).from_buffer_copy(piece)
            assert lib.jpeg_amd_stream_push(s, buf, len(piece), C.byref(scans), C.byref(fin)) == 0
        assert fin.value == 1 and scans.value == 1
        assert time.perf_counter() - t0 < 20.0
    finally:
        lib.jpeg_amd_stream_destroy(s)
