"""Host entropy encoder + file writer of libjpeg_amd.so -- CPU only.

The reference's own output files (examples/encode-basic/*.jpg, written by
Spectral.compress(stream:), encode.swift:1918) are the pins: entropy-decoding one of them and
encoding the coefficients again with the same scan structure must give the file back byte for
byte -- the optimised Huffman tables (heap tie-breaking, 16-bit limiting, symbol order), the
eager ZRL rule, the quantisation-slot allocation and the segment order are all in those bytes."""
import ctypes as C
import hashlib

import numpy as np
import pytest

import _golden as G
from jpeg_amd import _lib
from jpeg_amd.api import _scan_array, _jfif

# examples/encode-basic/main.swift: scan 1 = Y with tables (0, 0); scan 2 = Cb, Cr with (1, 1)
SCANS = [[(0, 0, 0)], [(1, 1, 1), (2, 1, 1)]]
JFIF = (2, 2, 1, 1)            # JPEG.JFIF(version: .v1_2, density: (1, 1, .centimeters))


def _decode(path):
    lib = _lib.lib()
    data = np.fromfile(path, np.uint8)
    info = _lib.FrameInfo()
    assert lib.jpeg_amd_jpeg_inspect(data.ctypes.data, data.size, C.byref(info)) == 0
    planes = [np.zeros((info.units_y[c], info.units_x[c], 64), np.int16) for c in range(info.ncomponents)]
    quanta = np.zeros((4, 64), np.uint16)
    assert lib.jpeg_amd_jpeg_decode_spectral(data.ctypes.data, data.size, _lib.ptr_array([p.ctypes.data for p in planes]),
                                             quanta.ctypes.data, None) == 0
    return data, info, planes, quanta


def _encode(info, planes, keys, tables, tkeys, scans=SCANS, jfif=JFIF):
    lib = _lib.lib()
    qkey = (C.c_int32 * len(keys))(*keys)
    tk = (C.c_int32 * len(tkeys))(*tkeys)
    tables = np.ascontiguousarray(tables, np.uint16)
    sarr = _scan_array(scans)
    j = _jfif(jfif)
    n = C.c_size_t()
    args = [C.byref(info), qkey, _lib.ptr_array([p.ctypes.data for p in planes]), tables.ctypes.data, tk, len(tkeys),
            sarr, len(scans), C.byref(j) if j is not None else None]
    st = lib.jpeg_amd_jpeg_encode_spectral(*args, None, 0, C.byref(n))
    if st != 0:
        return st, None
    out = np.empty(n.value, np.uint8)
    st = lib.jpeg_amd_jpeg_encode_spectral(*args, out.ctypes.data, out.size, C.byref(n))
    return st, out[:n.value]


@pytest.mark.parametrize("case", [c for c in G.encode_cases() if "file" in c], ids=lambda c: f"{c['mode']}-{c['level']}")
def test_reencoding_the_references_file_gives_it_back(case):
    data, info, planes, quanta = _decode(G.path(case["file"]))
    # components 1,2,3 use quanta keys 0,1,1 (main.swift:36-40)
    st, out = _encode(info, planes, [0, 1, 1], np.stack([quanta[0], quanta[1]]), [0, 1])
    assert st == 0
    assert out.size == data.size == case["file_nbytes"]
    assert (out == data).all()
    assert hashlib.sha256(out.tobytes()).hexdigest() == case["file_sha256"]


def test_round_trip_through_the_decoder_for_other_scan_structures():
    """One fully interleaved scan, and three separate scans with four-way table selectors
    (extended process): whatever the writer produces, the reader must give the planes back."""
    data, info, planes, quanta = _decode(G.path(next(c for c in G.encode_cases() if "file" in c)["file"]))
    for process, scans in ((0, [[(0, 0, 0), (1, 1, 1), (2, 1, 1)]]),
                           (1, [[(0, 0, 0)], [(1, 1, 2)], [(2, 3, 3)]])):
        info.process = process
        st, out = _encode(info, planes, [0, 1, 1], np.stack([quanta[0], quanta[1]]), [0, 1], scans=scans, jfif=None)
        assert st == 0
        lib = _lib.lib()
        info2 = _lib.FrameInfo()
        back = [np.full_like(p, 99) for p in planes]
        q2 = np.zeros((4, 64), np.uint16)
        assert lib.jpeg_amd_jpeg_decode_spectral(out.ctypes.data, out.size, _lib.ptr_array([p.ctypes.data for p in back]),
                                                 q2.ctypes.data, C.byref(info2)) == 0
        assert info2.nscans == len(scans) and info2.process == process
        for a, b in zip(planes, back):
            assert (a == b).all()
        assert (q2[:3] == quanta[:3]).all()


def test_long_zero_runs_follow_the_reference_rule():
    """encode.swift:938-956: ZRL is emitted when the 16th zero of a run arrives, so a block that
    ends in 16k zeros has no EOB and one ending in 16k + r zeros has k ZRLs and an EOB."""
    info = _lib.FrameInfo()
    info.width = info.height = 8
    info.precision, info.ncomponents, info.process = 8, 1, 0
    info.id[0], info.factor_x[0], info.factor_y[0], info.units_x[0], info.units_y[0] = 1, 1, 1, 1, 1
    q = np.ones((1, 64), np.uint16)
    for last in (63, 47, 31, 15, 40, 1):
        blk = np.zeros((1, 1, 64), np.int16)
        blk[0, 0, 0] = 5
        blk[0, 0, 1:last + 1] = -3          # no zero before `last`: only the trailing run counts
        st, out = _encode(info, [blk], [0], q, [0], scans=[[(0, 0, 0)]], jfif=None)
        assert st == 0
        back = np.zeros_like(blk)
        q2 = np.zeros((4, 64), np.uint16)
        assert _lib.lib().jpeg_amd_jpeg_decode_spectral(out.ctypes.data, out.size, _lib.ptr_array([back.ctypes.data]),
                                                        q2.ctypes.data, None) == 0
        assert (back == blk).all()
        # the DHT of the AC table lists ZRL (0xf0) exactly when 63 - last >= 16
        i = bytes(out).find(b"\xff\xc4")
        dht = bytes(out[i + 4:i + 2 + (out[i + 2] << 8 | out[i + 3])])
        ac = dht[dht.index(0x10, 17):]
        assert (0xf0 in ac[17:]) == (63 - last >= 16)
        assert (0x00 in ac[17:]) == ((63 - last) % 16 != 0)


def test_precondition_failures():
    data, info, planes, quanta = _decode(G.path(next(c for c in G.encode_cases() if "file" in c)["file"]))
    tables = np.stack([quanta[0], quanta[1]])
    assert _encode(info, planes, [0, 7, 1], tables, [0, 1])[0] == _lib.EINVAL          # missing quantization table
    assert _encode(info, planes, [0, 1, 1], tables, [0, 1], scans=[[(0, 2, 0)]])[0] == _lib.EINVAL   # baseline: selectors 0..1
    info.process = 2
    assert _encode(info, planes, [0, 1, 1], tables, [0, 1])[0] == _lib.ENOSUP           # progressive not written
