"""The encoder's table construction (jpeg_amd_huffman_build: encode.swift:597-760) on histograms skewed enough to need the
16-bit length limit, which none of the reference's own files exercises.  Two checks per histogram: the code is a valid
JPEG code (every symbol coded once, lengths 1 .. 16, the Kraft sum leaves room for the reserved all-ones word, lengths
never increase with frequency), and the digest over all tables equals tests/golden/huffman_limit.json -- recorded from the
round-5 build of this library (tests/golden/make_huffman_limit.py), whose construction restated the reference's limiter
level by level and reproduced all of its committed files byte for byte; round 6 re-expressed the limiter through the
Kraft sum (csrc/entropy_encode.cpp: limit_to_16_bits) and must not change a single table.  CPU only."""
import ctypes as C
import hashlib
import json
import os

import numpy as np

from jpeg_amd import _lib

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = json.load(open(os.path.join(HERE, "golden", "huffman_limit.json")))


def histograms(seed: int, count: int):
    """Seeded histograms of six kinds: uniform, tiny, exponential, Fibonacci, powers of two, mixed."""
    rng = np.random.default_rng(seed)
    fib = [1, 1]
    while len(fib) < 70:
        fib.append(fib[-1] + fib[-2])
    for it in range(count):
        f = np.zeros(256, dtype=np.int64)
        n = int(rng.integers(1, 257))
        syms = rng.integers(0, 256, size=n)
        kind = it % 6
        if kind == 0:
            vals = rng.integers(1, 1001, size=n)
        elif kind == 1:
            vals = rng.integers(1, 5, size=n)
        elif kind == 2:
            vals = 1 + np.minimum(4e15, np.exp2(rng.integers(0, 4000, size=n) / 100.0)).astype(np.int64)
        elif kind == 3:
            vals = np.array([fib[i] for i in rng.integers(0, 70, size=n)], dtype=np.int64)
        elif kind == 4:
            vals = np.left_shift(np.int64(1), rng.integers(0, 50, size=n))
        else:
            vals = np.where(rng.integers(0, 3, size=n) == 0, rng.integers(1, 4, size=n),
                            1 + np.left_shift(np.int64(1), rng.integers(0, 40, size=n)))
        f[syms] = vals
        yield f


def build(freq):
    counts = (C.c_uint8 * 16)()
    values = (C.c_uint8 * 256)()
    n = C.c_int32()
    st = _lib.lib().jpeg_amd_huffman_build(freq.ctypes.data_as(C.POINTER(C.c_int64)), counts, values, C.byref(n))
    assert st == 0, st
    return list(counts), list(values)[:n.value]


def test_limited_codes_are_valid_and_equal_the_recorded_tables():
    h = hashlib.sha256()
    limited = 0
    for f in histograms(GOLD["seed"], GOLD["count"]):
        counts, values = build(f)
        present = np.flatnonzero(f)
        assert sorted(values) == present.tolist()                      # every symbol of the histogram once
        assert sum(counts) == len(values)
        kraft = sum(c << (16 - (l + 1)) for l, c in enumerate(counts))
        assert kraft <= 65535                                           # room for the all-ones word (T.81 C.2)
        # shorter codes never go to rarer symbols: frequencies in code order do not increase
        fr = f[values]
        assert np.all(fr[:-1] >= fr[1:])
        limited += counts[15] > 0
        h.update(bytes(counts)); h.update(bytes(values))
    assert limited == GOLD["tables_with_16_bit_codes"]
    assert h.hexdigest() == GOLD["sha256"]
