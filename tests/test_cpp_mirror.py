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


@pytest.mark.gpu
def test_cpp_mirror_file_level_calls(tmp_path):
    """spectral::decompress / spectral::compress of the mirror: a reference-encoded file decodes
    to the oracle's pixels, compresses back to the SAME bytes, and the example's source picture
    compresses to the reference's own file (examples/encode-basic)."""
    import hashlib
    exe = _build(tmp_path)
    case = next(c for c in G.encode_cases() if c["mode"] == "4-2-0" and c["level"] == 1.0)
    src = G.path(case["file"])
    rgb, (w, h) = G.encode_source()
    raw = tmp_path / "src.rgb"
    raw.write_bytes(rgb.tobytes())
    prefix = str(tmp_path / "f")
    r = subprocess.run([exe, "--file", src, prefix, str(raw), str(w), str(h)], capture_output=True, text=True)
    assert r.returncode == 0 and "ok" in r.stdout, r.stderr
    from oracle import jpeg_reader
    img = jpeg_reader.read_jpeg(src)
    _, rect = O.decode(img.planes, img.quanta, img.factors, (img.width, img.height))
    assert (np.fromfile(prefix + ".rgb", np.uint8) == O.unpack_rgb8(rect, 3).reshape(-1)).all()
    assert open(prefix + ".jpg", "rb").read() == open(src, "rb").read()
    assert hashlib.sha256(open(prefix + ".enc.jpg", "rb").read()).hexdigest() == case["file_sha256"]
