"""The reference's own Huffman known-answer vectors (tests/unit/tests.swift:141-461, extracted as data into
tests/golden/huffman_unit.json by tests/golden/make_golden.py) against the host entropy coder's table code:
jpeg_amd_huffman_lookup mirrors JPEG.Table.Huffman.Decoder's subscript (decode.swift:1243-1261) -- including its
answer for a window that is no codeword: symbol 0, length 16 -- and jpeg_amd_huffman_build the encoder's table
construction (encode.swift:597-760).  CPU only."""
import ctypes as C
import json
import os

import numpy as np

from jpeg_amd import _lib

HERE = os.path.dirname(os.path.abspath(__file__))
VEC = json.load(open(os.path.join(HERE, "golden", "huffman_unit.json")))


def lookup(counts, values, window):
    c = (C.c_uint8 * 16)(*counts)
    v = (C.c_uint8 * max(1, len(values)))(*values)
    sym, length = C.c_int32(-1), C.c_int32(-1)
    st = _lib.lib().jpeg_amd_huffman_lookup(c, v, len(values), window, C.byref(sym), C.byref(length))
    assert st == 0, st
    return sym.value, length.value


def test_annex_k_ac_table_decodes_all_162_codewords():
    # tests/unit/tests.swift:141-361 (huffmanBuilding)
    t = VEC["annex_k_ac"]
    assert len(t["codewords"]) == 162
    for (length, code), want in zip(t["codewords"], t["symbols"]):
        assert lookup(t["counts"], t["values"], (code << (16 - length)) & 0xffff) == (want, length)
        # the bits behind the codeword do not matter
        assert lookup(t["counts"], t["values"], ((code << (16 - length)) | ((1 << (16 - length)) - 1)) & 0xffff) == (want, length)


def test_hand_made_streams_including_windows_that_are_no_codeword():
    # tests/unit/tests.swift:365-461 (huffmanCoding): the third stream contains two windows that match no codeword;
    # the reference yields a null symbol and skips 16 bits (decode.swift:1255-1258) instead of failing
    for case in VEC["coding"]:
        counts = [len(level) for level in case["levels"]]
        values = [s for level in case["levels"] for s in level]
        bits = case["bits"]
        padded = bits + "1" * 32          # JPEG.Bitstream pads with 1-bits (jpeg.swift:1888-1890)
        b, got = 0, []
        while b < len(bits):
            sym, length = lookup(counts, values, int(padded[b:b + 16], 2))
            got.append(sym)
            b += length
        assert got == case["symbols"]


def test_tables_the_reference_rejects_are_rejected():
    sym, length = C.c_int32(), C.c_int32()
    lib = _lib.lib()
    one = (C.c_uint8 * 3)(1, 2, 3)
    over = (C.c_uint8 * 16)(3, *([0] * 15))                   # three codes of length 1
    assert lib.jpeg_amd_huffman_lookup(over, one, 3, 0, C.byref(sym), C.byref(length)) == _lib.EINVAL
    short = (C.c_uint8 * 16)(1, 1, *([0] * 14))                # counts say 2 values, 3 are passed
    assert lib.jpeg_amd_huffman_lookup(short, one, 3, 0, C.byref(sym), C.byref(length)) == _lib.EINVAL


def canonical_codes(counts, values):
    code, k, out = 0, 0, {}
    for l in range(16):
        for _ in range(counts[l]):
            out[values[k]] = (code, l + 1)
            code += 1
            k += 1
        code <<= 1
    return out


def test_encoder_tables_round_trip_through_the_decoder():
    # tests/unit/tests.swift:462-510 (huffmanCodingSymmetric): symbols biased towards 128, table from their
    # frequencies, encode, decode, compare
    rng = np.random.default_rng(20240807)
    lib = _lib.lib()
    for n in (1, 2, 3, 10, 100, 1000, 20000):
        symbols = (rng.integers(0, 128, n) + rng.integers(0, 129, n)).astype(np.int64)
        freq = np.bincount(symbols, minlength=256).astype(np.int64)
        counts, values, nv = (C.c_uint8 * 16)(), (C.c_uint8 * 256)(), C.c_int32()
        assert lib.jpeg_amd_huffman_build(freq.ctypes.data, counts, values, C.byref(nv)) == 0
        counts, values = list(counts), list(values)[:nv.value]
        assert sorted(values) == sorted(np.flatnonzero(freq).tolist()) and sum(counts) == nv.value
        codes = canonical_codes(counts, values)
        assert all(length <= 16 for _, length in codes.values())
        assert all(code != (1 << length) - 1 for code, length in codes.values()), "the all-ones codeword is reserved"
        bits = "".join(format(codes[s][0], "0%db" % codes[s][1]) for s in symbols[:2000])
        padded, b, got = bits + "1" * 32, 0, []
        while b < len(bits):
            sym, length = lookup(counts, values, int(padded[b:b + 16], 2))
            got.append(sym)
            b += length
        assert got == symbols[:2000].tolist()
