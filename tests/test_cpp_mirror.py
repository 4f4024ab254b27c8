"""The C++ host mirror (include/jpeg_amd.hpp) over the C ABI.

CPU: the header and its driver compile and link against libjpeg_amd.so.
GPU: tests/cpp/host_mirror.cpp decodes a reference fixture staged and fused through the
mirror's Spectral / Planar / Rectangular classes and must reproduce the reference's gold
digests; the re-encoded coefficients must equal the oracle's."""
import os
import struct
import subprocess

import numpy as np
import pytest

import _golden as G
from oracle import oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _build(tmp_path):
    exe = str(tmp_path / "host_mirror")
    libdir = os.path.join(ROOT, "jpeg_amd")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", "host_mirror.cpp"), "-o", exe,
                           "-L", libdir, "-ljpeg_amd", "-Wl,-rpath," + libdir])
    return exe


def test_cpp_mirror_compiles_and_links(tmp_path):
    assert os.path.exists(_build(tmp_path))


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["color-sequential-1.jpg", "color-sequential-3.jpg", "grayscale-sequential-1.jpg"])
def test_cpp_mirror_reproduces_gold(tmp_path, name):
    exe = _build(tmp_path)
    img = G.image(name)
    n = len(img.components)
    blob = struct.pack("<3i", img.width, img.height, n)
    for i, c in enumerate(img.components):
        blob += struct.pack("<3i", c.fx, c.fy, i)
    blob += struct.pack("<i", n)
    for q in img.quanta:
        blob += np.asarray(q, np.uint16).tobytes()
    for p in img.planes:
        blob += np.ascontiguousarray(p, np.int16).tobytes()
    inp = tmp_path / "in.bin"
    inp.write_bytes(blob)
    prefix = str(tmp_path / "out")
    r = subprocess.run([exe, str(inp), prefix], capture_output=True, text=True)
    assert r.returncode == 0 and "ok" in r.stdout, r.stderr
    gold = G.entry(name)["gold"]
    rgb = np.fromfile(prefix + ".fused.rgb", np.uint8)
    assert G.sha(np.fromfile(prefix + ".staged.rgb", np.uint8)) == gold["rgb_sha256"]
    assert G.sha(rgb) == gold["rgb_sha256"]
    assert G.sha(np.fromfile(prefix + ".fused.ycc", np.uint8)) == gold["ycc_sha256"]
    # the mirror re-encoded the decoded picture with the same tables: compare with the oracle
    want = O.encode(rgb.reshape(-1, 3), (img.width, img.height), img.factors, img.quanta)
    for p, w in enumerate(want):
        got = np.fromfile(prefix + f".coef{p}", np.int16).reshape(w.shape)
        assert (got == w).all()
