"""Hostile and damaged input to the host entropy decoder: no C++ exception, abort or out-of-range read may come out
of the C ABI (include/jpeg_amd.h: "plain C, no exceptions, no aborts"); damaged entropy-coded data is rendered the way
the reference renders it (an unknown codeword is symbol 0 of length 16, decode.swift:1255-1258).  CPU only."""
import ctypes as C
import time

import numpy as np

import _golden as G
from jpeg_amd import _lib
from _sparse import sparse_decode, expand


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


def _decode(lib, data, info, careful, threads=1):
    planes = [np.zeros((info.units_y[c], info.units_x[c], 64), np.int16) for c in range(info.ncomponents)]
    quanta = np.zeros((4, 64), np.uint16)
    buf = (C.c_uint8 * len(data)).from_buffer_copy(bytes(data))
    # max_scans > 0 (a preview after N scans, here more than the file has) takes the careful bit reader: stuffing, markers
    # and the end of the data tested byte by byte, one symbol per lookup
    st = lib.jpeg_amd_jpeg_decode_spectral_partial(buf, len(data), _lib.ptr_array([p.ctypes.data for p in planes]),
                                                   quanta.ctypes.data, None, threads, 1000 if careful else 0)
    return st, planes


def test_fast_sequential_path_agrees_with_the_careful_reader_on_damaged_streams():
    # the fast path of sequential scans (stuffing removed up front, two AC symbols per lookup, DC + EOB in one) against the
    # byte-by-byte reader, on intact files and on the same files with flipped bits, garbage runs, stray 0xFF bytes and
    # markers, a truncated scan, and a removed restart marker (which sends both down the resynchronising path)
    lib = _lib.lib()
    rng = np.random.default_rng(2024)
    for name in ["color-sequential-1.jpg", "color-sequential-3.jpg", "color-sequential-restart.jpg",
                 "grayscale-sequential-1.jpg", "grayscale-sequential-restart.jpg", "karlie-kwk-2019.jpg"]:
        data = bytearray(fixture_bytes(name))
        info = _lib.FrameInfo()
        buf0 = (C.c_uint8 * len(data)).from_buffer_copy(bytes(data))
        assert lib.jpeg_amd_jpeg_inspect(buf0, len(data), C.byref(info)) == 0
        if info.process == 2:
            continue
        sos = data.index(b"\xff\xda")
        start = sos + 2 + int.from_bytes(data[sos + 2:sos + 4], "big")
        end = len(data) - 2
        for trial in range(40):
            d = bytearray(data)
            kind = trial % 8
            if kind == 1:                                   # a few flipped bits
                for _ in range(int(rng.integers(1, 6))):
                    d[int(rng.integers(start, end))] ^= 1 << int(rng.integers(8))
            elif kind == 2:                                 # a run of bytes without 0xFF
                lo = int(rng.integers(start, max(start + 1, end - 64)))
                d[lo:lo + 48] = bytes(rng.integers(0, 0xff, 48).astype(np.uint8).tolist())
            elif kind == 3:                                 # stray 0xFF bytes: stuffed, unstuffed, doubled
                lo = int(rng.integers(start, end - 4))
                d[lo:lo + 3] = [b"\xff\x00\xff", b"\xff\xff\x00", b"\xff\x01\x02"][int(rng.integers(3))]
            elif kind == 4:                                 # a marker in the middle of the data
                lo = int(rng.integers(start, end - 2))
                d[lo:lo + 2] = bytes([0xff, int(rng.choice([0xd0, 0xd3, 0xd7, 0xc4, 0xfe]))])
            elif kind == 5:                                 # truncated
                d = d[:int(rng.integers(start + 1, end))]
            elif kind == 6:                                 # one restart marker gone (if there is one)
                marks = [i for i in range(start, end - 1) if d[i] == 0xff and 0xd0 <= d[i + 1] <= 0xd7]
                if marks:
                    i = marks[int(rng.integers(len(marks)))]
                    d[i:i + 2] = b"\x12\x34"
            elif kind == 7:                                 # all ones to the end
                lo = int(rng.integers(start, end))
                d[lo:end] = b"\xfe" * (end - lo)
            a = _decode(lib, d, info, careful=False)
            b = _decode(lib, d, info, careful=True)
            assert a[0] == b[0], (name, trial, a[0], b[0])
            if a[0] == 0:
                for c, (p, q) in enumerate(zip(a[1], b[1])):
                    assert (p == q).all(), (name, trial, kind, c, np.argwhere(p != q)[:3])
            # the sparse output of the same decoder: the same coefficients as entries, or a refusal (ENOSUP: a marker is
            # missing or misplaced and the resynchronising reader, which writes planes, has to take the file)
            st, desc, ent, _q = sparse_decode(lib, d, info)
            assert st in (a[0], _lib.ENOSUP), (name, trial, st, a[0])
            if st == 0:
                for c, (p, q) in enumerate(zip(expand(info, desc, ent), a[1])):
                    assert (p == q).all(), (name, trial, kind, c, "sparse")
            if info.restart_interval and kind != 6:
                t = _decode(lib, d, info, careful=False, threads=3)
                assert t[0] == a[0] and all((p == q).all() for p, q in zip(t[1], a[1])), (name, trial)
