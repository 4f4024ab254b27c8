"""Pin the CPU oracle against every golden vector the reference holds for the hot path.

CPU only.  Fixtures + digests: tests/golden/MANIFEST.json (made by make_golden.py from
the reference's committed outputs).
"""
import numpy as np
import pytest

import _golden as G
from oracle import oracle as O


def _decode(name):
    img = G.image(name)
    n = len(img.components)
    planar, rect = O.decode(img.planes, img.quanta, img.factors, (img.width, img.height),
                            precision=img.precision)
    return img, n, planar, rect


@pytest.mark.parametrize("name", G.decode_names())
def test_reader_matches_manifest(name):
    img, e = G.image(name), G.entry(name)
    assert [img.width, img.height] == [e["width"], e["height"]]
    assert [list(f) for f in img.factors] == e["factors"]
    assert [G.sha(p) for p in img.planes] == e["coef_sha256"]


@pytest.mark.parametrize("name", G.decode_names(gold_only=True))
def test_decode_golds(name):
    """tests/regression/tests.swift:129 -- unpack(as: YCbCr) and unpack(as: RGB) equal gold."""
    img, n, planar, rect = _decode(name)
    gold = G.entry(name)["gold"]
    if "ycc_sha256" in gold:
        ycc = O.unpack_ycc8(rect, n)
        assert ycc.nbytes == gold["ycc_nbytes"]
        assert G.sha(ycc) == gold["ycc_sha256"]
    rgb = O.unpack_rgb8(rect, n)
    assert rgb.nbytes == gold["rgb_nbytes"]
    assert G.sha(rgb) == gold["rgb_sha256"]
    if "rgb_file" in gold:  # full dump kept for one image: compare element-wise too
        g = np.fromfile(G.path(gold["rgb_file"]), np.uint8).reshape(-1, 3)
        assert (rgb == g).all()


def test_idct_stage_planes():
    """examples/decode-advanced: the only per-stage pin of Spectral.idct()."""
    img, n, planar, rect = _decode("karlie-2019.jpg")
    for p, g in enumerate(G.entry("karlie-2019.jpg")["gold"]["planes"]):
        want = np.fromfile(G.path(g["file"]), np.uint8)
        w, h = map(int, g["dims"].split("x"))
        assert planar[p].shape == (h, w)
        assert (planar[p].astype(np.uint8).reshape(-1) == want).all()
        assert G.sha(want) == g["sha256"]


@pytest.mark.parametrize("case", G.encode_cases(), ids=lambda c: f"{c['mode']}-{c['level']}")
def test_encode_golds(case):
    """pack + decomposed + fdct == coefficients inside the reference's encode-basic JPEGs."""
    rgb, size = G.encode_source()
    factors = [tuple(f) for f in case["factors"]]
    ql = O.compression_quanta("luminance", case["level"])
    qc = O.compression_quanta("chrominance", case["level"])
    assert [q.tolist() for q in (ql, qc, qc)] == case["quanta_zigzag"]
    planes = O.encode(rgb, size, factors, [ql, qc, qc])
    assert [[p.shape[1], p.shape[0]] for p in planes] == case["units"]
    assert [G.sha(p) for p in planes] == case["coef_sha256"]


@pytest.mark.parametrize("case", [c for c in G.encode_cases() if "file" in c],
                         ids=lambda c: f"{c['mode']}-{c['level']}")
def test_encode_golds_against_files(case):
    """Same, but re-reading the committed reference JPEGs (not just their digests)."""
    from oracle import jpeg_reader
    img = jpeg_reader.read_jpeg(G.path(case["file"]))
    rgb, size = G.encode_source()
    planes = O.encode(rgb, size, [tuple(f) for f in case["factors"]], img.quanta)
    for a, b in zip(planes, img.planes):
        assert (a == b).all()


def test_twelve_bit_four_component_encode_pin():
    """examples/custom-color (main.swift:132-200): the reference compressed a deterministic RGBA12 gradient into
    output.jpg -- 12-bit, four components, factors (2,2) x 3 + (1,1), 16-bit tables -- and dumped that input beside it.
    decomposed() (a (1,1) plane inside a (2,2) scale: 2 x 2 box filter, encode.swift:389-425) + load(limit:) + fdct at
    P = 12 (encode.swift:80-99, 199-248) on the dump == the coefficients inside the file: the pin SURVEY 8c thought
    missing for precision 12."""
    from oracle import jpeg_reader
    values, size, factors, quanta, file, m = G.custom_color()
    img = jpeg_reader.read_jpeg(file)
    assert (img.precision, len(img.planes)) == (12, 4) and m["precision"] == 12
    assert [q.tolist() for q in quanta] == [img.quanta[i].tolist() for i in range(4)]
    planes = O.decompose(values, size, factors, (2, 2))
    for p, q, want, digest in zip(planes, quanta, img.planes, m["coef_sha256"]):
        got = O.fdct_plane(p, q, 12)
        assert got.shape == want.shape and (got == want).all()
        assert G.sha(got) == digest
