"""GPU parity: the HIP path (through the C ABI) against the CPU oracle and the golden
vectors.  Bit-exact on every output integer -- tolerance 0."""
import ctypes as C

import numpy as np
import pytest

import _golden as G
from oracle import oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def J():
    import jpeg_amd
    return jpeg_amd


@pytest.fixture(scope="module")
def ctx(J):
    return J.Context(0)


def _layout_for(J, img):
    n = len(img.components)
    fmt = "y8" if n == 1 else "ycc8"
    comps = {c.ident: J.Component((c.fx, c.fy), i) for i, c in enumerate(img.components)}
    return J.Layout(fmt, comps)


def _spectral(J, ctx, img):
    layout = _layout_for(J, img)
    return J.Spectral.from_host(ctx, (img.width, img.height), layout, img.planes, img.quanta,
                                q=list(range(len(img.components))))


# ---- config 2: 100k synthetic blocks, IDCT + dequant only ---------------------------------

@pytest.mark.parametrize("dist", ["uniform", "natural"])
@pytest.mark.parametrize("table", ["ones", "luminance1.0"])
def test_c2_100k_blocks_idct(J, ctx, dist, table):
    from jpeg_amd import synth
    blocks = (synth.blocks_uniform if dist == "uniform" else synth.blocks_natural)(100_000)
    coef = blocks.reshape(250, 400, 64)
    q = np.ones(64, np.uint16) if table == "ones" else J.compression_quanta("luminance", 1.0)
    layout = J.Layout("y8", {1: J.Component((1, 1), 0)})
    spectral = J.Spectral.from_host(ctx, (3200, 2000), layout, [coef], [q])
    got = spectral.idct().host_planes()[0]
    want = O.idct_plane(coef, q, 8, threads=8)
    assert got.shape == want.shape
    assert (got == want).all(), f"{(got != want).sum()} of {got.size} samples differ"


# ---- staged decode on every fixture ---------------------------------------------------------

@pytest.mark.parametrize("name", G.decode_names())
def test_staged_decode_matches_oracle_and_gold(J, ctx, name):
    img = G.image(name)
    n = len(img.components)
    spectral = _spectral(J, ctx, img)
    planar_o, rect_o = O.decode(img.planes, img.quanta, img.factors, (img.width, img.height))

    planar = spectral.idct()
    for got, want in zip(planar.host_planes(), planar_o):
        assert (got == want).all()
    rect = planar.interleaved(cosite=False)
    assert (rect.host_values() == rect_o).all()
    ycc = rect.unpack(J.YCbCr).cpu().numpy()
    rgb = rect.unpack(J.RGB).cpu().numpy()
    assert (ycc == O.unpack_ycc8(rect_o, n)).all()
    assert (rgb == O.unpack_rgb8(rect_o, n)).all()
    gold = G.entry(name)["gold"]
    if "ycc_sha256" in gold:
        assert G.sha(ycc) == gold["ycc_sha256"]
    if "rgb_sha256" in gold:
        assert G.sha(rgb) == gold["rgb_sha256"]


@pytest.mark.parametrize("name", G.decode_names())
@pytest.mark.parametrize("color", ["rgb", "ycc"])
def test_fused_decode_matches_gold(J, ctx, name, color):
    img = G.image(name)
    n = len(img.components)
    spectral = _spectral(J, ctx, img)
    got = spectral.decode(J.RGB if color == "rgb" else J.YCbCr).cpu().numpy()
    _, rect_o = O.decode(img.planes, img.quanta, img.factors, (img.width, img.height))
    want = (O.unpack_rgb8 if color == "rgb" else O.unpack_ycc8)(rect_o, n)
    assert (got == want).all(), f"{(got != want).sum()} bytes differ"
    gold = G.entry(name)["gold"]
    if color + "_sha256" in gold:
        assert G.sha(got) == gold[color + "_sha256"]


def test_idct_stage_planes_match_decode_advanced_dump(J, ctx):
    img = G.image("karlie-2019.jpg")
    planes = _spectral(J, ctx, img).idct().host_planes()
    for p, g in enumerate(G.entry("karlie-2019.jpg")["gold"]["planes"]):
        want = np.fromfile(G.path(g["file"]), np.uint8)
        assert (planes[p].astype(np.uint8).reshape(-1) == want).all()


# ---- cosited + unusual layouts (no gold in the reference: oracle is the checker) ------------

LAYOUTS = [
    # (size, [(factor, qi)...] recognised, extra non-recognised factor or None, precision)
    ((1, 1), [((1, 1), 0)], None, 8),
    ((7, 9), [((2, 2), 0), ((1, 1), 1), ((1, 1), 1)], None, 8),
    ((17, 33), [((2, 1), 0), ((1, 1), 1), ((1, 1), 1)], None, 8),
    ((33, 17), [((1, 2), 0), ((1, 1), 1), ((1, 1), 1)], None, 8),
    ((50, 41), [((4, 1), 0), ((1, 1), 1), ((1, 1), 1)], None, 8),     # 4:1:1
    ((45, 37), [((3, 2), 0), ((1, 1), 1), ((3, 1), 1)], None, 8),     # odd factors
    ((40, 24), [((1, 1), 0), ((1, 1), 1), ((1, 1), 1)], (2, 2), 8),   # scale from a non-recognised comp
    ((37, 29), [((2, 2), 0), ((1, 1), 1), ((1, 1), 1), ((2, 1), 0)], None, 12),  # custom 4-plane 12-bit
]


def _random_spectral(J, ctx, size, comps, extra, precision, seed):
    rng = np.random.default_rng(seed)
    keyed = {i + 1: J.Component(f, qi) for i, (f, qi) in enumerate(comps)}
    if extra is not None:
        keyed[99] = J.Component(extra, 0)
    n = len(comps)
    fmt = "y8" if (n == 1 and precision == 8) else "ycc8" if (n == 3 and precision == 8) else ("custom", precision, n)
    layout = J.Layout(fmt, keyed)
    units = layout.units(size)
    amp = 1 << (precision + 1)
    planes = []
    for ux, uy in units:
        c = rng.integers(-amp, amp, (uy, ux, 64)).astype(np.int16)
        c[..., 8:] //= 8
        planes.append(c)
    nq = max(qi for _, qi in comps) + 1
    quanta = [rng.integers(1, 40, 64).astype(np.uint16) for _ in range(nq)]
    q = [qi for _, qi in comps]
    return layout, planes, quanta, q


@pytest.mark.parametrize("case", range(len(LAYOUTS)))
@pytest.mark.parametrize("cosite", [False, True])
def test_staged_decode_unusual_layouts(J, ctx, case, cosite):
    size, comps, extra, precision = LAYOUTS[case]
    layout, planes, quanta, q = _random_spectral(J, ctx, size, comps, extra, precision, 100 + case)
    spectral = J.Spectral.from_host(ctx, size, layout, planes, quanta, q=q)
    planar = spectral.idct()
    factors = [c.factor for c in layout.planes]
    planar_o = [O.idct_plane(p, quanta[qi], precision) for p, qi in zip(planes, q)]
    for got, want in zip(planar.host_planes(), planar_o):
        assert (got == want).all()
    rect = planar.interleaved(cosite=cosite).host_values()
    want = O.interleave(planar_o, factors, layout.scale, size, cosited=cosite)
    assert (rect == want).all(), f"{(rect != want).sum()} samples differ"
    # the same in one call (jpeg_amd_spectral_rectangular: k_generic_fused where every factor is 1 | 2, staged otherwise)
    assert (spectral.rectangular(cosite=cosite).host_values() == want).all()
    if precision == 8 and len(comps) in (1, 3):
        for color, unpack in ((J.RGB, O.unpack_rgb8), (J.YCbCr, O.unpack_ycc8)):
            got = spectral.decode(color, cosite=cosite).cpu().numpy()
            assert (got == unpack(want, len(comps))).all()


# ---- encode ------------------------------------------------------------------------------------

def _encode_layout(J, factors):
    return J.Layout("ycc8", {1: J.Component(tuple(factors[0]), 0),
                             2: J.Component(tuple(factors[1]), 1),
                             3: J.Component(tuple(factors[2]), 1)})


@pytest.mark.parametrize("case", G.encode_cases(), ids=lambda c: f"{c['mode']}-{c['level']}")
def test_encode_matches_reference_coefficients(J, ctx, case):
    rgb, size = G.encode_source()
    layout = _encode_layout(J, case["factors"])
    quanta = {0: J.compression_quanta("luminance", case["level"]),
              1: J.compression_quanta("chrominance", case["level"])}
    # staged: pack -> decomposed -> fdct
    rect = J.Rectangular.pack(ctx, size, layout, rgb, J.RGB)
    spectral = rect.decomposed().fdct(quanta)
    assert [G.sha(p) for p in spectral.host_planes()] == case["coef_sha256"]
    # fused
    fused = J.Rectangular.encode(ctx, size, layout, rgb, quanta, J.RGB)
    assert [G.sha(p) for p in fused.host_planes()] == case["coef_sha256"]


def test_encode_stages_match_oracle(J, ctx):
    rgb, size = G.encode_source()
    factors = [(2, 2), (1, 1), (1, 1)]
    layout = _encode_layout(J, factors)
    rect = J.Rectangular.pack(ctx, size, layout, rgb, J.RGB)
    want_rect = O.pack_rgb8(rgb, 3).reshape(size[1], size[0], 3)
    assert (rect.host_values() == want_rect).all()
    planar = rect.decomposed()
    want_planes = O.decompose(want_rect, size, factors, (2, 2))
    for got, want in zip(planar.host_planes(), want_planes):
        assert (got == want).all()
    # YCbCr pack is a widening copy
    ycc = O.unpack_ycc8(want_rect, 3)
    rect2 = J.Rectangular.pack(ctx, size, layout, ycc, J.YCbCr)
    assert (rect2.host_values() == want_rect).all()


@pytest.mark.parametrize("case", [1, 2, 4, 5])
def test_encode_unusual_layouts(J, ctx, case):
    size, comps, extra, precision = LAYOUTS[case]
    rng = np.random.default_rng(7 + case)
    rgb = rng.integers(0, 256, (size[0] * size[1], 3)).astype(np.uint8)
    layout = J.Layout("ycc8", {i + 1: J.Component(f, qi) for i, (f, qi) in enumerate(comps)})
    factors = [c.factor for c in layout.planes]
    quanta = {0: rng.integers(1, 30, 64).astype(np.uint16), 1: rng.integers(1, 30, 64).astype(np.uint16)}
    want = O.encode(rgb, size, factors, [quanta[qi] for _, qi in comps], scale=layout.scale)
    got = J.Rectangular.pack(ctx, size, layout, rgb, J.RGB).decomposed().fdct(quanta).host_planes()
    for a, b in zip(got, want):
        assert (a == b).all()
    fused = J.Rectangular.encode(ctx, size, layout, rgb, quanta, J.RGB).host_planes()
    for a, b in zip(fused, want):
        assert (a == b).all()


def test_fdct_idct_roundtrip_property(J, ctx):
    """encode -> decode round trip with all-ones quanta reproduces the image within 1 LSB
    per YCbCr sample (size-independent property; no oracle involved)."""
    from jpeg_amd import synth
    size = (256, 192)
    rgb = synth.smooth_rgb(*size)
    layout = _encode_layout(J, [(1, 1), (1, 1), (1, 1)])
    ones = {0: np.ones(64, np.uint16), 1: np.ones(64, np.uint16)}
    spectral = J.Rectangular.encode(ctx, size, layout, rgb, ones, J.RGB)
    ycc_back = spectral.decode(J.YCbCr).cpu().numpy().astype(int)
    ycc_in = J.Rectangular.pack(ctx, size, layout, rgb, J.RGB).unpack(J.YCbCr).cpu().numpy().astype(int)
    assert np.abs(ycc_back - ycc_in).max() <= 1


# ---- error behaviour + host-buffer ABI ---------------------------------------------------------

def test_precondition_failures_come_back_as_einval(J, ctx):
    from jpeg_amd import _lib
    layout = J.Layout("ycc8", {1: J.Component((1, 1), 0), 2: J.Component((1, 1), 0), 3: J.Component((1, 1), 0)})
    with pytest.raises(J.JpegAmdError) as e:   # decode.swift:1710 array count does not match
        J.Rectangular.from_host(ctx, (4, 4), layout, np.zeros(5, np.uint16))
    assert e.value.status == _lib.EINVAL
    planar = J.Planar.from_host(ctx, (8, 8), layout, [np.zeros((8, 8), np.uint16)] * 3)
    with pytest.raises(J.JpegAmdError):        # decode.swift:2527 missing quantization table
        planar.fdct({7: np.ones(64, np.uint16)})
    L = layout.c_layout((0, 8))
    st = _lib.lib().jpeg_amd_planar_interleaved(ctx.handle, C.byref(L), _lib.ptr_array([1, 1, 1]), 0, 1)
    assert st == _lib.EINVAL                   # decode.swift:2599 size must be positive
    # subsampled planes too small for the image: the bilinear sample index of the last pixel would fall outside the
    # plane (the reference traps on the array access, decode.swift:4243-4257); EINVAL here, never an out-of-bounds read
    sub = J.Layout("ycc8", {1: J.Component((2, 2), 0), 2: J.Component((1, 1), 0), 3: J.Component((1, 1), 0)})
    L = sub.c_layout((64, 64), [(8, 8), (1, 1), (1, 1)], [0, 0, 0])   # chroma planes of 8 x 8 samples for 64 x 64 pixels
    st = _lib.lib().jpeg_amd_planar_interleaved(ctx.handle, C.byref(L), _lib.ptr_array([1, 1, 1]), 0, 1)
    assert st == _lib.EINVAL
    st = _lib.lib().jpeg_amd_planar_interleaved(ctx.handle, C.byref(L), _lib.ptr_array([1, 1, 1]), 1, 1)
    assert st == _lib.EINVAL


def test_host_buffer_abi(J, ctx):
    """jpeg_amd_host_* (what the Swift shim binds): host pointers in, host pointers out."""
    from jpeg_amd import _lib
    img = G.image("color-sequential-1.jpg")
    layout = _layout_for(J, img)
    size = (img.width, img.height)
    L = layout.c_layout(size, [(c.ux, c.uy) for c in img.components], [0, 1, 2])
    q = np.ascontiguousarray(np.stack(img.quanta).astype(np.uint16))
    coef = [np.ascontiguousarray(p) for p in img.planes]
    out = np.empty((img.width * img.height, 3), np.uint8)
    lib = _lib.lib()
    st = lib.jpeg_amd_host_decode(ctx.handle, C.byref(L), _lib.ptr_array([c.ctypes.data for c in coef]),
                                  q.ctypes.data, 3, 0, _lib.COLOR_RGB8, out.ctypes.data)
    assert st == 0
    assert G.sha(out) == G.entry("color-sequential-1.jpg")["gold"]["rgb_sha256"]

    planes = [np.empty((8 * c.uy, 8 * c.ux), np.uint16) for c in img.components]
    st = lib.jpeg_amd_host_spectral_idct(ctx.handle, C.byref(L), _lib.ptr_array([c.ctypes.data for c in coef]),
                                         q.ctypes.data, 3, _lib.ptr_array([p.ctypes.data for p in planes]))
    assert st == 0
    rect = np.empty((img.height, img.width, 3), np.uint16)
    st = lib.jpeg_amd_host_planar_interleaved(ctx.handle, C.byref(L), _lib.ptr_array([p.ctypes.data for p in planes]),
                                              0, rect.ctypes.data)
    assert st == 0
    ycc = np.empty((img.width * img.height, 3), np.uint8)
    st = lib.jpeg_amd_host_rectangular_unpack(ctx.handle, rect.ctypes.data, img.width * img.height, 3,
                                              _lib.COLOR_YCC8, ycc.ctypes.data)
    assert st == 0
    assert G.sha(ycc) == G.entry("color-sequential-1.jpg")["gold"]["ycc_sha256"]
    # idct().interleaved() in one call with host buffers (what Spectral.rectangular(cosite:) of the shim binds)
    rect2 = np.empty((img.height, img.width, 3), np.uint16)
    st = lib.jpeg_amd_host_spectral_rectangular(ctx.handle, C.byref(L), _lib.ptr_array([c.ctypes.data for c in coef]),
                                                q.ctypes.data, 3, 0, rect2.ctypes.data)
    assert st == 0 and (rect2 == rect).all()

    # encode side
    rgb, esize = G.encode_source()
    case = next(c for c in G.encode_cases() if c["mode"] == "4-2-0" and c["level"] == 1.0)
    elay = _encode_layout(J, case["factors"])
    EL = elay.c_layout(esize, None, [0, 1, 1])
    eq = np.ascontiguousarray(np.stack([J.compression_quanta("luminance", 1.0),
                                        J.compression_quanta("chrominance", 1.0)]))
    ecoef = [np.empty((uy, ux, 64), np.int16) for ux, uy in elay.units(esize)]
    st = lib.jpeg_amd_host_encode(ctx.handle, C.byref(EL), np.ascontiguousarray(rgb).ctypes.data, _lib.COLOR_RGB8,
                                  eq.ctypes.data, 2, _lib.ptr_array([c.ctypes.data for c in ecoef]))
    assert st == 0
    assert [G.sha(c) for c in ecoef] == case["coef_sha256"]
    # staged host encode
    erect = np.empty((esize[1], esize[0], 3), np.uint16)
    assert lib.jpeg_amd_host_rectangular_pack(ctx.handle, np.ascontiguousarray(rgb).ctypes.data, esize[0] * esize[1], 3,
                                              _lib.COLOR_RGB8, erect.ctypes.data) == 0
    eplanes = [np.empty((8 * uy, 8 * ux), np.uint16) for ux, uy in elay.units(esize)]
    assert lib.jpeg_amd_host_rectangular_decomposed(ctx.handle, C.byref(EL), erect.ctypes.data,
                                                    _lib.ptr_array([p.ctypes.data for p in eplanes])) == 0
    ecoef2 = [np.empty((uy, ux, 64), np.int16) for ux, uy in elay.units(esize)]
    assert lib.jpeg_amd_host_planar_fdct(ctx.handle, C.byref(EL), _lib.ptr_array([p.ctypes.data for p in eplanes]),
                                         eq.ctypes.data, 2, _lib.ptr_array([c.ctypes.data for c in ecoef2])) == 0
    assert [G.sha(c) for c in ecoef2] == case["coef_sha256"]


# ---- seeded sweep over frame geometry: every edge path of the fused kernels -----------------

def _sweep_cases(n=48):
    rng = np.random.default_rng(77)
    cases = []
    for i in range(n):
        w = int(rng.choice([rng.integers(1, 40), rng.integers(40, 700), 16 * rng.integers(1, 40), 256 * rng.integers(1, 3) + rng.integers(-3, 4)]))
        h = int(rng.choice([rng.integers(1, 40), rng.integers(40, 300), 16 * rng.integers(1, 12) + rng.integers(-2, 3)]))
        fac = [(1, 1), (2, 1), (1, 2), (2, 2)][int(rng.integers(4))]
        cases.append((max(w, 1), max(h, 1), fac, bool(rng.integers(2)), int(rng.integers(3))))
    return cases


@pytest.mark.parametrize("case", _sweep_cases(), ids=lambda c: f"{c[0]}x{c[1]}-{c[2][0]}{c[2][1]}-{'rgb' if c[3] else 'ycc'}-{c[4]}")
def test_fused_kernels_on_random_geometry(J, ctx, case):
    """Fused decode and fused encode against the oracle for sizes that hit partial strips,
    partial tiles, rows that are not 16-byte multiples (byte-wise store tail), single blocks,
    and grey images (mode 2)."""
    w, h, fac, rgb, mode = case
    comps = [((1, 1), 0)] if mode == 2 else [(fac, 0), ((1, 1), 1), ((1, 1), 1)]
    layout, planes, quanta, q = _random_spectral(J, ctx, (w, h), comps, None, 8, 1000 + w * 7 + h)
    factors = [c.factor for c in layout.planes]
    spectral = J.Spectral.from_host(ctx, (w, h), layout, planes, quanta, q=q)
    color, unpack = (J.RGB, O.unpack_rgb8) if rgb else (J.YCbCr, O.unpack_ycc8)
    got = spectral.decode(color).cpu().numpy()
    _, rect = O.decode(planes, [quanta[i] for i in q], factors, (w, h))
    want = unpack(rect, len(comps))
    assert (got == want).all(), f"decode: {(got != want).sum()} bytes differ"
    # encode the decoded picture again
    qmap = {i: quanta[i] for i in set(q)}
    coef = J.Rectangular.encode(ctx, (w, h), layout, want, qmap, color).host_planes()
    pack = O.pack_rgb8 if rgb else O.pack_ycc8
    planar = O.decompose(pack(want, len(comps)).reshape(h, w, len(comps)), (w, h), factors, layout.scale)
    enc = [O.fdct_plane(p, quanta[i]) for p, i in zip(planar, q)]
    for a, b in zip(coef, enc):
        assert (a == b).all(), "encode differs"


# 4:2:2: the neighbour blocks' edge columns are transformed by (block, column) work-items (idct_edge_col_split, dct.hpp): a
# strip column without a left neighbour, without a right one, with both, a plane that ends inside the strip, two block rows
# of which the second lies below the plane, extreme coefficients (the clamp), and every strip column of a wide image.
@pytest.mark.parametrize("size", [(256, 16), (512, 16), (1024, 48), (272, 16), (768, 24), (4096, 32), (520, 40), (248, 8)])
@pytest.mark.parametrize("rgb", [True, False])
def test_422_edge_columns_by_block_and_column(J, ctx, size, rgb):
    w, h = size
    comps = [((2, 1), 0), ((1, 1), 1), ((1, 1), 1)]
    layout, planes, quanta, q = _random_spectral(J, ctx, (w, h), comps, None, 8, 4220 + w + h)
    rng = np.random.default_rng(w * 31 + h)
    for p in planes[1:]:   # chroma: a third of the blocks with coefficients that drive the samples past both clamps
        hot = rng.random(p.shape[:2]) < 0.33
        p[hot] = rng.integers(-1024, 1024, (int(hot.sum()), 64)).astype(np.int16)
    factors = [c.factor for c in layout.planes]
    spectral = J.Spectral.from_host(ctx, (w, h), layout, planes, quanta, q=q)
    color, unpack = (J.RGB, O.unpack_rgb8) if rgb else (J.YCbCr, O.unpack_ycc8)
    got = spectral.decode(color).cpu().numpy()
    _, rect = O.decode(planes, [quanta[i] for i in q], factors, (w, h))
    want = unpack(rect, 3)
    assert (got == want).all(), f"{(got != want).sum()} bytes differ"


# ---- f-4: generic JPEG.Format plug-ins.  The ENCODE half at precision 12 / four planes is pinned on a
# ---- reference-held vector: examples/custom-color's output.jpg + the dump of its input (tests/test_oracle_golden.py::
# ---- test_twelve_bit_four_component_encode_pin for the oracle, tests/test_gpu_compress.py::test_twelve_bit_... for the
# ---- device).  The DECODE half (12 / 16-bit idct, cosited and odd-factor upsampling) has no gold in the reference: there
# ---- the oracle is the only checker -- the tests below say so in their names ------------------------------------------

@pytest.mark.parametrize("precision", [12, 16])
@pytest.mark.parametrize("nplanes", [1, 3, 4])
def test_generic_format_decode_unpinned_no_reference_gold_and_encode_beyond_the_pinned_case(J, ctx, precision, nplanes):
    """Spectral.idct / Planar.interleaved (centred and cosited) and Rectangular.decomposed / Planar.fdct for custom
    formats of 12 and 16 bits: decode.swift:4101-4133 with level 2^(P-1) + 0.5 and clamp to 2^P - 1, and
    encode.swift:80-99 where load() clamps samples ABOVE the format's limit (min(limit, sample)) -- the samples fed
    to the encoder here exceed it on purpose."""
    rng = np.random.default_rng(1000 * precision + nplanes)
    factors = [(2, 2), (1, 1), (1, 2), (2, 1)][:nplanes] if nplanes > 1 else [(1, 1)]
    size = (61, 45)
    layout = J.Layout(("custom", precision, nplanes), {i + 1: J.Component(f, i & 1) for i, f in enumerate(factors)})
    units = layout.units(size)
    amp = 1 << (precision - 1)
    planes = []
    for ux, uy in units:
        c = rng.integers(-amp, amp, (uy, ux, 64)).astype(np.int32)
        c[..., 5:] //= 32
        planes.append(np.clip(c, -32768, 32767).astype(np.int16))
    tables = [rng.integers(1, 12, 64).astype(np.uint16) for _ in range(2)]
    q = [c.qi for c in layout.planes]
    spectral = J.Spectral.from_host(ctx, size, layout, planes, tables, q=q)
    planar = spectral.idct()
    want_p = [O.idct_plane(p, tables[i], precision) for p, i in zip(planes, q)]
    for got, want in zip(planar.host_planes(), want_p):
        assert (got == want).all()
    assert max(int(w.max()) for w in want_p) == (1 << precision) - 1      # the clamp at 2^P - 1 is exercised
    fs = [c.factor for c in layout.planes]
    for cosite in (False, True):
        rect = planar.interleaved(cosite=cosite).host_values()
        assert (rect == O.interleave(want_p, fs, layout.scale, size, cosited=cosite)).all()
        assert (spectral.rectangular(cosite=cosite).host_values() == rect).all()   # one launch (k_generic_fused), no Planar in HBM
    # encode: samples over the whole uint16 range, i.e. above the limit for P = 12
    samples = rng.integers(0, 65536, (size[1], size[0], nplanes)).astype(np.uint16)
    back = J.Rectangular.from_host(ctx, size, layout, samples).decomposed()
    want_d = O.decompose(samples, size, fs, layout.scale)
    for got, want in zip(back.host_planes(), want_d):
        assert (got == want).all()
    # (quanta >= 16: with random 16-bit samples smaller divisors would push coefficients past Int16, where the
    # reference traps)
    enc_tables = [rng.integers(16, 60, 64).astype(np.uint16) for _ in range(2)]
    sp2 = back.fdct({i: enc_tables[i] for i in range(2)})
    want_f = [O.fdct_plane(p, enc_tables[c.qi], precision) for p, c in zip(want_d, layout.planes)]
    assert max(int(np.abs(w.astype(np.int32)).max()) for w in want_f) < 32000
    for got, want in zip(sp2.host_planes(), want_f):
        assert (got == want).all()
    if precision == 12:
        over = [O.fdct_plane(np.minimum(p, 4095), enc_tables[c.qi], precision) for p, c in zip(want_d, layout.planes)]
        assert all((a == b).all() for a, b in zip(want_f, over)), "oracle: load(limit:) clamps at 2^P - 1"
        assert any((p > 4095).any() for p in want_d)


@pytest.mark.parametrize("seed", range(4))
def test_generic_format_seeded_sweep_decode_unpinned_no_reference_gold(J, ctx, seed):
    """Ten random formats per seed (1-4 planes, factors 1-4, 8 / 12 / 16 bits, centred or cosited) through the staged
    kernels both ways -- the collected share of tests/soak_staged.py."""
    rng = np.random.default_rng(9000 + seed)
    for _ in range(10):
        n = int(rng.integers(1, 5))
        precision = int(rng.choice([8, 12, 16]))
        w, h = int(rng.integers(1, 120)), int(rng.integers(1, 90))
        comps = {i + 1: J.Component((int(rng.integers(1, 5)), int(rng.integers(1, 5))), int(rng.integers(0, 2))) for i in range(n)}
        layout = J.Layout(("custom", precision, n), comps)
        units = layout.units((w, h))
        amp = 1 << (precision + 1)
        planes = []
        for ux, uy in units:
            c = rng.integers(-amp, amp, (uy, ux, 64)).astype(np.int32)
            c[..., 6:] //= 16
            planes.append(np.clip(c, -32768, 32767).astype(np.int16))
        quanta = [rng.integers(1, 50, 64).astype(np.uint16) for _ in range(2)]
        q = [c.qi for c in layout.planes]
        keys = sorted(set(q)); q = [keys.index(k) for k in q]; tables = [quanta[k] for k in keys]
        cosite = bool(rng.integers(2))
        tag = (w, h, n, precision, cosite, [c.factor for c in layout.planes])
        spectral = J.Spectral.from_host(ctx, (w, h), layout, planes, tables, q=q)
        planar = spectral.idct()
        want_p = [O.idct_plane(p, tables[i], precision) for p, i in zip(planes, q)]
        assert all((a == b).all() for a, b in zip(planar.host_planes(), want_p)), tag
        factors = [c.factor for c in layout.planes]
        want_r = O.interleave(want_p, factors, layout.scale, (w, h), cosited=cosite)
        assert (planar.interleaved(cosite=cosite).host_values() == want_r).all(), tag
        assert (spectral.rectangular(cosite=cosite).host_values() == want_r).all(), tag
        back = J.Rectangular.from_host(ctx, (w, h), layout, want_r).decomposed()
        want_d = O.decompose(want_r.reshape(h, w, n), (w, h), factors, layout.scale)
        assert all((a == b).all() for a, b in zip(back.host_planes(), want_d)), tag
        sp2 = back.fdct({c.qi: quanta[c.qi] for c in layout.planes})
        want_f = [O.fdct_plane(p, quanta[c.qi], precision) for p, c in zip(want_d, layout.planes)]
        assert all((a == b).all() for a, b in zip(sp2.host_planes(), want_f)), tag


@pytest.mark.parametrize("sampling", [[(2, 2), (1, 1), (1, 1)], [(2, 1), (1, 1), (1, 1)], [(1, 2), (1, 1), (1, 1)], [(1, 1), (1, 1), (1, 1)]])
def test_cosited_through_the_fused_entry_point_parity_unpinned_no_reference_gold(J, ctx, sampling):
    """jpeg_amd_decode(cosited = 1) on the built-in ycc8 format: the fused kernels only implement centred upsampling,
    so the call must take the staged kernels and still agree with Planar.interleaved(cosite: true)
    (decode.swift:4223-4230) + unpack.  The reference has no cosited gold."""
    rng = np.random.default_rng(31)
    size = (203, 97)
    layout = J.Layout("ycc8", {i + 1: J.Component(f, min(i, 1)) for i, f in enumerate(sampling)})
    units = layout.units(size)
    planes = [np.clip(rng.laplace(0, 40, (uy, ux, 64)), -1000, 1000).astype(np.int16) for ux, uy in units]
    tables = [rng.integers(1, 20, 64).astype(np.uint16) for _ in range(2)]
    q = [c.qi for c in layout.planes]
    spectral = J.Spectral.from_host(ctx, size, layout, planes, tables, q=q)
    want_p = [O.idct_plane(p, tables[i], 8) for p, i in zip(planes, q)]
    fs = [c.factor for c in layout.planes]
    for cosite in (True, False):
        rect = O.interleave(want_p, fs, layout.scale, size, cosited=cosite)
        assert (spectral.decode(J.RGB, cosite=cosite).cpu().numpy() == O.unpack_rgb8(rect, 3)).all()
        assert (spectral.decode(J.YCbCr, cosite=cosite).cpu().numpy() == O.unpack_ycc8(rect, 3)).all()



GENERIC_FUSED = [
    # (size, precision, factors, cosite): layouts k_generic_fused takes (every factor 1 | 2 under a scale <= 2), sizes of several tiles
    ((700, 333), 12, [(2, 2), (1, 1), (1, 1)], False),
    ((700, 333), 12, [(2, 2), (1, 1), (1, 1)], True),
    ((513, 129), 12, [(2, 2), (2, 2), (2, 2), (1, 1)], False),     # examples/custom-color/main.swift:150-158: rgba12
    ((385, 200), 16, [(1, 1), (1, 1), (1, 1)], False),
    ((300, 301), 12, [(2, 1), (1, 1), (1, 1)], True),
    ((257, 190), 8, [(1, 2), (1, 1), (1, 2)], False),
    ((129, 65), 9, [(1, 1), (1, 1)], False),                       # two planes under a scale set by a non-recognised component
    ((1000, 64), 12, [(1, 1)], False),
    ((128, 64), 12, [(2, 2), (1, 1), (1, 1)], False),              # exactly one tile
    ((1, 1), 16, [(2, 2), (1, 1), (1, 2), (2, 1)], True),
]


@pytest.mark.parametrize("case", range(len(GENERIC_FUSED)))
def test_generic_fused_decode_unpinned_no_reference_gold_matches_oracle_and_staged(J, ctx, case):
    """jpeg_amd_spectral_rectangular == idct().interleaved(cosite:) (decode.swift:4154-4165, 4182-4276) for custom formats, one launch:
    against the oracle AND against the staged kernels, on images of several tiles, every plane mix, extreme coefficients (the
    clamp at 2^P - 1 is reached).  12 / 16-bit and cosited decodes have no gold in the reference (examples/custom-color dumps
    its INPUT): the oracle is the only checker, as for the staged path."""
    size, precision, factors, cosite = GENERIC_FUSED[case]
    rng = np.random.default_rng(4200 + case)
    n = len(factors)
    comps = {i + 1: J.Component(f, i & 1) for i, f in enumerate(factors)}
    if case == 6:
        comps[99] = J.Component((2, 2), 0)
    layout = J.Layout(("custom", precision, n), comps)
    units = layout.units(size)
    amp = 1 << (precision - 1)
    planes = []
    for ux, uy in units:
        c = rng.integers(-amp, amp, (uy, ux, 64)).astype(np.int32)
        c[..., 5:] //= 32
        planes.append(np.clip(c, -32768, 32767).astype(np.int16))
    tables = [rng.integers(1, 12, 64).astype(np.uint16) for _ in range(2)]
    q = [c.qi for c in layout.planes]
    keys = sorted(set(q)); q = [keys.index(k) for k in q]; tables = [tables[k] for k in keys]
    spectral = J.Spectral.from_host(ctx, size, layout, planes, tables, q=q)
    want_p = [O.idct_plane(p, tables[i], precision) for p, i in zip(planes, q)]
    want = O.interleave(want_p, [c.factor for c in layout.planes], layout.scale, size, cosited=cosite)
    got = spectral.rectangular(cosite=cosite).host_values()
    assert (got == want).all(), f"{(got != want).sum()} of {want.size} samples differ from the oracle"
    assert (spectral.idct().interleaved(cosite=cosite).host_values() == got).all()
    if size[0] * size[1] > 1:
        assert int(want.max()) == (1 << precision) - 1 and int(want.min()) == 0   # both clamps exercised


def test_generic_fused_batch_strides(J, ctx):
    """jpeg_amd_spectral_rectangular_batch: three images of one 12-bit 4:2:0 layout, per-image table sets, strides as in
    jpeg_amd_decode_batch; every image equals its own single call."""
    import ctypes as C
    import torch
    from jpeg_amd import _lib
    rng = np.random.default_rng(77)
    size, n = (210, 130), 3
    layout = J.Layout(("custom", 12, 3), {1: J.Component((2, 2), 0), 2: J.Component((1, 1), 1), 3: J.Component((1, 1), 1)})
    units = layout.units(size)
    L = layout.c_layout(size, units, [0, 1, 1])
    host = [[np.clip(rng.laplace(0, 300, (uy, ux, 64)), -8000, 8000).astype(np.int16) for ux, uy in units] for _ in range(n)]
    tables = rng.integers(1, 30, (n, 2, 64)).astype(np.uint16)
    d_planes = [torch.from_numpy(np.stack([host[i][p] for i in range(n)])).to(ctx.torch_device) for p in range(3)]
    d_q = torch.from_numpy(tables.view(np.int16)).to(ctx.torch_device)
    out = torch.zeros((n, size[0] * size[1] * 3), dtype=torch.int16, device=ctx.torch_device)
    st = _lib.lib().jpeg_amd_spectral_rectangular_batch(ctx.handle, C.byref(L), n, _lib.ptr_array([p.data_ptr() for p in d_planes]),
                                                        _lib.size_array([64 * a * b for a, b in units]), d_q.data_ptr(), 128, 2, 0,
                                                        out.data_ptr(), size[0] * size[1] * 3)
    assert st == 0
    got = out.cpu().numpy().view(np.uint16)
    for i in range(n):
        want_p = [O.idct_plane(host[i][p], tables[i][min(p, 1)], 12) for p in range(3)]
        want = O.interleave(want_p, [(2, 2), (1, 1), (1, 1)], (2, 2), size, cosited=False)
        assert (got[i] == want.reshape(-1)).all(), i


@pytest.mark.parametrize("cosited", [False, True])
def test_generic_fused_tile_walk_over_several_rounds_and_images(J, ctx, cosited):
    """The resident, ticket-drawn tile walk of k_generic_fused (4:2:0 layouts): five 12-bit images of 2000 x 1490 with their own
    tables are 5 x 16 x 24 = 1 920 tiles -- more than the 1 024 workgroups resident at once, so tiles are drawn from the counter, across
    image boundaries (the tables change under a resident workgroup), with edge tiles in both directions; twice in a row (the last
    workgroup to leave must have put the counter back).  Against the oracle on one image and against each image's own single call."""
    import ctypes as C
    import torch
    from jpeg_amd import _lib
    rng = np.random.default_rng(4100 + cosited)
    size, n = (2000, 1490), 5
    layout = J.Layout(("custom", 12, 3), {1: J.Component((2, 2), 0), 2: J.Component((1, 1), 1), 3: J.Component((1, 1), 1)})
    units = layout.units(size)
    L = layout.c_layout(size, units, [0, 1, 1])
    host = [[np.clip(rng.laplace(0, 300, (uy, ux, 64)), -8000, 8000).astype(np.int16) for ux, uy in units] for _ in range(n)]
    tables = rng.integers(1, 30, (n, 2, 64)).astype(np.uint16)
    d_planes = [torch.from_numpy(np.stack([host[i][p] for i in range(n)])).to(ctx.torch_device) for p in range(3)]
    d_q = torch.from_numpy(tables.view(np.int16)).to(ctx.torch_device)
    npx = size[0] * size[1] * 3
    lib = _lib.lib()
    outs = []
    for _ in range(2):
        out = torch.zeros((n, npx), dtype=torch.int16, device=ctx.torch_device)
        st = lib.jpeg_amd_spectral_rectangular_batch(ctx.handle, C.byref(L), n, _lib.ptr_array([p.data_ptr() for p in d_planes]),
                                                     _lib.size_array([64 * a * b for a, b in units]), d_q.data_ptr(), 128, 2, int(cosited),
                                                     out.data_ptr(), npx)
        assert st == 0
        outs.append(out)
    assert torch.equal(outs[0], outs[1])
    single = torch.zeros(npx, dtype=torch.int16, device=ctx.torch_device)
    for i in range(n):
        st = lib.jpeg_amd_spectral_rectangular_batch(ctx.handle, C.byref(L), 1, _lib.ptr_array([p[i].data_ptr() for p in d_planes]),
                                                     _lib.size_array([0, 0, 0]), d_q[i].data_ptr(), 0, 2, int(cosited), single.data_ptr(), 0)
        assert st == 0
        assert torch.equal(single, outs[0][i]), i
    i = 3
    want_p = [O.idct_plane(host[i][p], tables[i][min(p, 1)], 12) for p in range(3)]
    want = O.interleave(want_p, [(2, 2), (1, 1), (1, 1)], (2, 2), size, cosited=cosited)
    assert (outs[0][i].cpu().numpy().view(np.uint16) == want.reshape(-1)).all()


@pytest.mark.parametrize("case", range(len(GENERIC_FUSED)))
def test_generic_fused_encode_matches_oracle_and_staged(J, ctx, case):
    """jpeg_amd_rectangular_spectral == decomposed().fdct(quanta:) (encode.swift:389-425, 199-248) for custom formats, one launch:
    against the oracle AND against the staged kernels, on the layouts and sizes of the decode test; samples over the whole range
    of the precision (the min(limit, .) of encode.swift:85 included: a few samples lie above 2^P - 1)."""
    size, precision, factors, _cosite = GENERIC_FUSED[case]
    rng = np.random.default_rng(5200 + case)
    n = len(factors)
    comps = {i + 1: J.Component(f, i & 1) for i, f in enumerate(factors)}
    if case == 6:
        comps[99] = J.Component((2, 2), 0)
    layout = J.Layout(("custom", precision, n), comps)
    w, h = size
    top = (1 << precision) - 1
    yy, xx = np.mgrid[0:h, 0:w]
    base = (0.5 + 0.45 * np.sin(xx / 9.0) * np.cos(yy / 7.0))[..., None] * top
    values = np.clip(base + rng.integers(-top // 8 - 1, top // 8 + 2, (h, w, n)), 0, 65535).astype(np.uint16)
    values[rng.integers(0, h, 5), rng.integers(0, w, 5)] = min(65535, top + 3)          # above the limit of the precision
    quanta = {0: rng.integers(1, 40, 64).astype(np.uint16), 1: rng.integers(1, 40, 64).astype(np.uint16)}
    rect = J.Rectangular.from_host(ctx, size, layout, values)
    got = rect.spectral(quanta).host_planes()
    flist = [c.factor for c in layout.planes]
    want_d = O.decompose(values, size, flist, layout.scale)
    want = [O.fdct_plane(p, quanta[c.qi], precision) for p, c in zip(want_d, layout.planes)]
    for p, (a, b) in enumerate(zip(got, want)):
        assert (a == b).all(), f"plane {p}: {(a != b).sum()} of {b.size} coefficients differ from the oracle"
    staged = rect.decomposed().fdct(quanta).host_planes()
    assert all((a == b).all() for a, b in zip(got, staged))


def test_generic_fused_encode_batch_strides_and_host_buffers(J, ctx):
    """jpeg_amd_rectangular_spectral_batch: three images of one 12-bit 4:2:0 layout, per-image table sets, strides; every image
    equals the oracle; jpeg_amd_host_rectangular_spectral gives the same planes from host buffers."""
    import ctypes as C
    import torch
    from jpeg_amd import _lib
    rng = np.random.default_rng(78)
    size, n = (210, 130), 3
    layout = J.Layout(("custom", 12, 3), {1: J.Component((2, 2), 0), 2: J.Component((1, 1), 1), 3: J.Component((1, 1), 1)})
    units = layout.units(size)
    L = layout.c_layout(size, units, [0, 1, 1])
    values = rng.integers(0, 4096, (n, size[1], size[0], 3)).astype(np.uint16)
    tables = rng.integers(1, 30, (n, 2, 64)).astype(np.uint16)
    pad = 40
    d_rect = torch.zeros((n, size[0] * size[1] * 3 + pad), dtype=torch.int16, device=ctx.torch_device)
    d_rect[:, :size[0] * size[1] * 3] = torch.from_numpy(values.reshape(n, -1).view(np.int16)).to(ctx.torch_device)
    d_q = torch.from_numpy(tables.view(np.int16)).to(ctx.torch_device)
    out = [torch.full((n, 64 * a * b + 64), 99, dtype=torch.int16, device=ctx.torch_device) for a, b in units]
    st = _lib.lib().jpeg_amd_rectangular_spectral_batch(ctx.handle, C.byref(L), n, d_rect.data_ptr(), size[0] * size[1] * 3 + pad,
                                                        d_q.data_ptr(), 128, 2, _lib.ptr_array([o.data_ptr() for o in out]),
                                                        _lib.size_array([64 * a * b + 64 for a, b in units]))
    assert st == 0
    factors = [(2, 2), (1, 1), (1, 1)]
    for i in range(n):
        want_d = O.decompose(values[i], size, factors, (2, 2))
        for p in range(3):
            want = O.fdct_plane(want_d[p], tables[i][min(p, 1)], 12)
            got = out[p][i].cpu().numpy()
            assert (got[:want.size] == want.reshape(-1)).all(), (i, p)
            assert (got[want.size:] == 99).all()
    h_out = [np.zeros((b, a, 64), np.int16) for a, b in units]
    st = _lib.lib().jpeg_amd_host_rectangular_spectral(ctx.handle, C.byref(L), values[1].ctypes.data, tables[1].ctypes.data, 2,
                                                       _lib.ptr_array([o.ctypes.data for o in h_out]))
    assert st == 0
    for p in range(3):
        assert (h_out[p].reshape(-1) == out[p][1].cpu().numpy()[:h_out[p].size]).all()
