"""jpeg_amd_compress: pixels -> JPEG file bytes (GPU spectral path + host entropy encoder,
SURVEY.md 8f-3).  The pins are the reference's own 32 output files of examples/encode-basic
(4 subsampling modes x 8 quality levels), by SHA-256 and, for the 8 committed ones, byte by byte."""
import ctypes as C
import hashlib

import numpy as np
import pytest

import _golden as G
from jpeg_amd import _lib
from jpeg_amd.api import _scan_array, _metadata_array

pytestmark = pytest.mark.gpu

SCANS = [[(0, 0, 0)], [(1, 1, 1), (2, 1, 1)]]      # examples/encode-basic/main.swift:42-46
JFIF = [("jfif", (2, 2, 1, 1))]                                # .init(version: .v1_2, density: (1, 1, .centimeters))


@pytest.fixture(scope="module")
def ctx():
    import jpeg_amd as J
    return J.Context()


def _quanta(level):
    import jpeg_amd as J
    return np.stack([J.compression_quanta("luminance", level), J.compression_quanta("chrominance", level)]).astype(np.uint16)


@pytest.mark.parametrize("case", G.encode_cases(), ids=lambda c: f"{c['mode']}-{c['level']}")
def test_compress_reproduces_the_references_file(ctx, case):
    import jpeg_amd as J
    rgb, (w, h) = G.encode_source()
    info = _lib.FrameInfo()
    info.width, info.height, info.precision, info.ncomponents, info.process = w, h, 8, 3, 0
    for c, (fx, fy) in enumerate(case["factors"]):
        info.id[c], info.factor_x[c], info.factor_y[c] = c + 1, fx, fy
    tables = _quanta(case["level"])
    assert [t.tolist() for t in tables] == case["quanta_zigzag"][:2]
    qkey = (C.c_int32 * 3)(0, 1, 1)
    tk = (C.c_int32 * 2)(0, 1)
    sarr, n = _scan_array(SCANS), C.c_size_t()
    marr, nmeta, _keep = _metadata_array(JFIF)
    out = np.empty(1 << 20, np.uint8)
    px = np.ascontiguousarray(rgb)
    st = _lib.lib().jpeg_amd_compress(ctx.handle, C.byref(info), px.ctypes.data, J.RGB.code, qkey, tables.ctypes.data,
                                      tk, 2, sarr, 2, marr, nmeta, out.ctypes.data, out.size, C.byref(n))
    assert st == 0, st
    assert [[info.units_x[c], info.units_y[c]] for c in range(3)] == case["units"]
    got = out[:n.value]
    assert n.value == case["file_nbytes"]
    assert hashlib.sha256(got.tobytes()).hexdigest() == case["file_sha256"]
    if "file" in case:
        assert (got == np.fromfile(G.path(case["file"]), np.uint8)).all()


@pytest.mark.parametrize("mode", ["4-2-0", "4-4-4"])
def test_python_mirror_compress_then_decompress(ctx, mode, tmp_path):
    """Rectangular.pack(...).compress(path:quanta:) then Rectangular.decompress(path:) -- the
    reference's own round trip (examples/encode-basic + decode-basic)."""
    import jpeg_amd as J
    case = next(c for c in G.encode_cases() if c["mode"] == mode and c["level"] == 1.0)
    rgb, size = G.encode_source()
    layout = J.Layout("ycc8", {1: (tuple(case["factors"][0]), 0), 2: ((1, 1), 1), 3: ((1, 1), 1)})
    quanta = {0: J.compression_quanta("luminance", 1.0), 1: J.compression_quanta("chrominance", 1.0)}
    path = str(tmp_path / "out.jpg")
    data = J.Rectangular.pack(ctx, size, layout, rgb, J.RGB).compress(quanta, SCANS, metadata=JFIF, path=path)
    assert hashlib.sha256(data).hexdigest() == case["file_sha256"]
    back = J.Rectangular.decompress(ctx, path).unpack(J.RGB).cpu().numpy()
    err = back.astype(np.int32) - rgb.astype(np.int32)
    psnr = 10 * np.log10(255.0 ** 2 / np.mean(err.astype(np.float64) ** 2))
    assert psnr > 25.0, psnr


def test_progressive_compress_round_trip(ctx, tmp_path):
    """examples/encode-advanced's progression (successive approximation, refinement scans, a
    comment record) on the GPU-encoded coefficients: the written file must decode back to exactly
    those coefficients (through this library's own progressive decoder, which is pinned by the
    reference's progressive golds)."""
    import jpeg_amd as J
    from jpeg_amd.api import Scan
    rgb, size = G.encode_source()
    layout = J.Layout("ycc8", {1: ((2, 1), 0), 2: ((1, 1), 1), 3: ((1, 1), 1)})
    quanta = {0: [1, 2, 2, 3, 3, 3] + [4] * 58, 1: [1, 2, 2, 5, 5, 5] + [30] * 58}
    Y, Cb, Cr = 0, 1, 2
    scans = [Scan.progressive_dc((Y, 0), (Cb, 1), (Cr, 1), bits=2),
             Scan.progressive_dc_refine(Y, Cb, Cr, bit=1), Scan.progressive_dc_refine(Y, Cb, Cr, bit=0),
             Scan.progressive_ac((Y, 0), (1, 64), bits=1),
             Scan.progressive_ac((Cb, 0), (1, 6), bits=1), Scan.progressive_ac((Cr, 0), (1, 6), bits=1),
             Scan.progressive_ac((Cb, 0), (6, 64), bits=1), Scan.progressive_ac((Cr, 0), (6, 64), bits=1),
             Scan.progressive_ac_refine((Y, 0), (1, 64), bit=0),
             Scan.progressive_ac_refine((Cb, 0), (1, 64), bit=0), Scan.progressive_ac_refine((Cr, 0), (1, 64), bit=0)]
    spectral = J.Rectangular.pack(ctx, size, layout, rgb, J.RGB).decomposed().fdct(quanta)
    data = spectral.compress(scans, process="progressive", metadata=[("comment", b"the way u say 'important' is important")])
    back = J.Spectral.decompress(ctx, data)
    assert J.inspect(data).nscans == 11 and J.inspect(data).process == 2
    for a, b in zip(spectral.host_planes(), back.host_planes()):
        assert (a == b).all()
    assert all((np.asarray(x) == np.asarray(y)).all() for x, y in zip(back.quanta, [quanta[0], quanta[1], quanta[1]]))


def test_compress_batch_writes_the_references_file_for_every_image(ctx):
    """jpeg_amd_compress_batch: 5 copies of the example picture in one call (host threads do the
    entropy coding) -- every file is the reference's, and a buffer that is too small is reported."""
    import jpeg_amd as J
    case = next(c for c in G.encode_cases() if c["mode"] == "4-2-0" and c["level"] == 0.5)
    rgb, (w, h) = G.encode_source()
    n = 5
    px = np.ascontiguousarray(np.tile(rgb.reshape(1, -1), (n, 1)))
    info = _lib.FrameInfo()
    info.width, info.height, info.precision, info.ncomponents, info.process = w, h, 8, 3, 0
    for c, (fx, fy) in enumerate(case["factors"]):
        info.id[c], info.factor_x[c], info.factor_y[c] = c + 1, fx, fy
    tables = _quanta(case["level"])
    qkey, tk = (C.c_int32 * 3)(0, 1, 1), (C.c_int32 * 2)(0, 1)
    sarr = _scan_array(SCANS)
    marr, nmeta, _keep = _metadata_array(JFIF)
    cap = 1 << 18
    out = np.zeros((n, cap), np.uint8)
    sizes = (C.c_size_t * n)()
    st = _lib.lib().jpeg_amd_compress_batch(ctx.handle, C.byref(info), px.ctypes.data, 0, n, J.RGB.code, qkey, tables.ctypes.data,
                                            tk, 2, sarr, 2, marr, nmeta, 3, out.ctypes.data, cap, sizes)
    assert st == 0, st
    for i in range(n):
        assert sizes[i] == case["file_nbytes"]
        assert hashlib.sha256(out[i, :sizes[i]].tobytes()).hexdigest() == case["file_sha256"]
    st = _lib.lib().jpeg_amd_compress_batch(ctx.handle, C.byref(info), px.ctypes.data, 0, n, J.RGB.code, qkey, tables.ctypes.data,
                                            tk, 2, sarr, 2, marr, nmeta, 3, out.ctypes.data, 1000, sizes)
    assert st == _lib.EINVAL and sizes[0] == case["file_nbytes"]


def test_twelve_bit_four_component_compress_reproduces_the_references_file(ctx):
    """examples/custom-color/output.jpg from the gradient the reference made it from (tests/golden/make_golden.py keeps
    the reference's dump of that input): Rectangular.decomposed() -> Planar.fdct(quanta:) on the device at precision 12
    == the coefficient planes inside the file, and compress(...) with the example's scan progression (main.swift:101-118)
    == the file, byte for byte.  (The CPU twin, oracle against the same file: tests/test_oracle_golden.py.)"""
    import jpeg_amd as J
    from oracle import jpeg_reader
    from test_entropy_encode_cpu import _script
    values, size, factors, quanta, file, m = G.custom_color()
    data = np.fromfile(file, np.uint8)
    process, metadata, scans, keys, tkeys, tables = _script(data)
    assert process == 2 and keys == [0, 0, 0, 1]
    layout = J.Layout(("custom", 12, 4), {i: J.Component(f, k) for i, f, k in zip(m["idents"], factors, keys)})
    rect = J.Rectangular.from_host(ctx, size, layout, values)
    qd = {k: tables[tkeys.index(k)] for k in tkeys}
    spectral = rect.decomposed().fdct(qd)
    want = jpeg_reader.read_jpeg(file).planes
    for got, w in zip(spectral.host_planes(), want):
        assert got.shape == w.shape and (got == w).all()
    out = rect.compress(qd, scans, process="progressive", metadata=metadata)
    assert bytes(out) == data.tobytes()


@pytest.mark.parametrize("pinned", [False, True])
def test_compress_batch_over_several_chunks_pageable_and_pinned_pixels(ctx, pinned):
    """70 pictures = three chunks of the pipeline; the pixels either in pageable memory (staged through the library's pinned
    slots by the host threads) or page-locked (uploaded from where they are), with a pitch between the pictures; picture i
    is the example picture with its first bytes changed, so that a chunk or slot mix-up cannot go unnoticed."""
    import torch
    import jpeg_amd as J
    case = next(c for c in G.encode_cases() if c["mode"] == "4-2-0" and c["level"] == 0.5)
    rgb, (w, h) = G.encode_source()
    n = 70
    stride = w * h * 3 + 32
    holder = torch.zeros(n * stride, dtype=torch.uint8, pin_memory=pinned)
    px = holder.numpy()
    for i in range(n):
        px[i * stride:i * stride + w * h * 3] = rgb.reshape(-1)
        px[i * stride:i * stride + 24] = (i * 3) & 255
    info = _lib.FrameInfo()
    info.width, info.height, info.precision, info.ncomponents, info.process = w, h, 8, 3, 0
    for c, (fx, fy) in enumerate(case["factors"]):
        info.id[c], info.factor_x[c], info.factor_y[c] = c + 1, fx, fy
    tables = _quanta(case["level"])
    qkey, tk = (C.c_int32 * 3)(0, 1, 1), (C.c_int32 * 2)(0, 1)
    sarr = _scan_array(SCANS)
    marr, nmeta, _keep = _metadata_array(JFIF)
    cap = 1 << 18
    lib = _lib.lib()
    want = {}
    for threads in (1, 6):
        out = np.zeros((n, cap), np.uint8)
        sizes = (C.c_size_t * n)()
        st = lib.jpeg_amd_compress_batch(ctx.handle, C.byref(info), px.ctypes.data, stride, n, J.RGB.code, qkey, tables.ctypes.data,
                                         tk, 2, sarr, 2, marr, nmeta, threads, out.ctypes.data, cap, sizes)
        assert st == 0, st
        for i in range(n):
            got = out[i, :sizes[i]].tobytes()
            if i not in want:      # the single-picture entry point on the same pixels
                one = np.zeros(cap, np.uint8)
                nb = C.c_size_t()
                f1 = _lib.FrameInfo()
                C.memmove(C.byref(f1), C.byref(info), C.sizeof(f1))
                st = lib.jpeg_amd_compress_batch(ctx.handle, C.byref(f1), px[i * stride:].ctypes.data, 0, 1, J.RGB.code, qkey,
                                                 tables.ctypes.data, tk, 2, sarr, 2, marr, nmeta, 1, one.ctypes.data, cap, C.byref(nb))
                assert st == 0, st
                want[i] = one[:nb.value].tobytes()
            assert got == want[i], (i, threads)
    assert len(set(want.values())) > 40     # the pictures do differ


def test_compress_batch_device_takes_the_pixels_where_they_are(ctx):
    """jpeg_amd_compress_batch_device: 40 pictures already in device memory (two chunks), with a pitch: the same files as
    jpeg_amd_compress_batch writes from host memory."""
    import torch
    import jpeg_amd as J
    case = next(c for c in G.encode_cases() if c["mode"] == "4-2-0" and c["level"] == 0.5)
    rgb, (w, h) = G.encode_source()
    n = 40
    stride = w * h * 3 + 16
    px = np.zeros(n * stride, np.uint8)
    for i in range(n):
        px[i * stride:i * stride + w * h * 3] = rgb.reshape(-1)
        px[i * stride:i * stride + 30] = (i * 5) & 255
    d_px = torch.from_numpy(px).to(ctx.torch_device)
    info = _lib.FrameInfo()
    info.width, info.height, info.precision, info.ncomponents, info.process = w, h, 8, 3, 0
    for c, (fx, fy) in enumerate(case["factors"]):
        info.id[c], info.factor_x[c], info.factor_y[c] = c + 1, fx, fy
    tables = _quanta(case["level"])
    qkey, tk = (C.c_int32 * 3)(0, 1, 1), (C.c_int32 * 2)(0, 1)
    sarr = _scan_array(SCANS)
    marr, nmeta, _keep = _metadata_array(JFIF)
    cap = 1 << 18
    lib = _lib.lib()
    outs = []
    for fn, src in ((lib.jpeg_amd_compress_batch, px.ctypes.data), (lib.jpeg_amd_compress_batch_device, d_px.data_ptr())):
        out = np.zeros((n, cap), np.uint8)
        sizes = (C.c_size_t * n)()
        st = fn(ctx.handle, C.byref(info), src, stride, n, J.RGB.code, qkey, tables.ctypes.data, tk, 2, sarr, 2, marr, nmeta, 4,
                out.ctypes.data, cap, sizes)
        assert st == 0, st
        outs.append([out[i, :sizes[i]].tobytes() for i in range(n)])
    assert outs[0] == outs[1]
    assert len(set(outs[0])) > 20


def test_compress_batch_pictures_too_dense_for_the_sparse_download_come_down_as_planes(ctx):
    """Noise under all-ones tables has ~64 nonzero coefficients per block: more than the 24 per block a picture's arena holds,
    so its planes come down whole while the smooth picture between two noisy ones comes down as entries -- every file equals
    what the staged mirror (jpeg_amd_encode + jpeg_amd_jpeg_encode_spectral on the planes) writes."""
    import jpeg_amd as J
    rng = np.random.default_rng(99)
    w, h, n = 200, 136, 5
    yy, xx = np.mgrid[0:h, 0:w]
    smooth = np.clip(128 + 60 * np.sin(xx / 31.0) * np.cos(yy / 17.0), 0, 255).astype(np.uint8)
    px = np.zeros((n, h, w, 3), np.uint8)
    for i in range(n):
        px[i] = rng.integers(0, 256, (h, w, 3)) if i % 2 == 0 else smooth[..., None]
    layout = J.Layout("ycc8", {1: ((2, 2), 0), 2: ((1, 1), 1), 3: ((1, 1), 1)})
    ones = {0: np.ones(64, np.uint16), 1: np.ones(64, np.uint16)}
    want = [J.Rectangular.pack(ctx, (w, h), layout, px[i].reshape(-1, 3), J.RGB).decomposed().fdct(ones).compress(SCANS, metadata=JFIF) for i in range(n)]
    info = _lib.FrameInfo()
    info.width, info.height, info.precision, info.ncomponents, info.process = w, h, 8, 3, 0
    for c, (fx, fy) in enumerate([(2, 2), (1, 1), (1, 1)]):
        info.id[c], info.factor_x[c], info.factor_y[c] = c + 1, fx, fy
    tables = np.ones((2, 64), np.uint16)
    qkey, tk = (C.c_int32 * 3)(0, 1, 1), (C.c_int32 * 2)(0, 1)
    sarr = _scan_array(SCANS)
    marr, nmeta, _keep = _metadata_array(JFIF)
    cap = 1 << 18
    out = np.zeros((n, cap), np.uint8)
    sizes = (C.c_size_t * n)()
    st = _lib.lib().jpeg_amd_compress_batch(ctx.handle, C.byref(info), px.ctypes.data, 0, n, J.RGB.code, qkey, tables.ctypes.data,
                                            tk, 2, sarr, 2, marr, nmeta, 3, out.ctypes.data, cap, sizes)
    assert st == 0, st
    for i in range(n):
        assert out[i, :sizes[i]].tobytes() == want[i], i
    assert sizes[0] > 4 * sizes[1]          # the noisy pictures really are dense


def test_custom_format_file_round_trip_through_one_c_call_each_way(ctx):
    """The boundary for custom formats (VERDICT r05 missing 1): Rectangular<Format>.compress(stream:quanta:) and
    Rectangular<Format>.decompress(stream:cosite:) (encode.swift:2031, decode.swift:4367-4374; examples/custom-color) as ONE
    call of the C ABI each, host memory in and out.  jpeg_amd_compress_rectangular on the gradient the reference made
    examples/custom-color/output.jpg from == that file, byte for byte (12 bits, four components, 16-bit DQT, progressive);
    jpeg_amd_decompress_rectangular of the file == the oracle's idct() + interleaved() of the file's planes (the decode of
    12-bit data has no gold in the reference: the oracle is the checker there), and == the chained calls of the Python mirror."""
    import jpeg_amd as J
    from oracle import jpeg_reader, oracle as O
    from test_entropy_encode_cpu import _script
    values, size, factors, quanta, file, m = G.custom_color()
    data = np.fromfile(file, np.uint8)
    process, metadata, scans, keys, tkeys, tables = _script(data)
    layout = J.Layout(("custom", 12, 4), {i: J.Component(f, k) for i, f, k in zip(m["idents"], factors, keys)})
    qd = {k: tables[tkeys.index(k)] for k in tkeys}
    out = J.Rectangular.compress_from_host(ctx, size, layout, values, qd, scans, process="progressive", metadata=metadata)
    assert bytes(out) == data.tobytes()

    info, rect = J.Rectangular.decompress_to_host(ctx, data.tobytes())
    assert (info.width, info.height, info.precision, info.ncomponents) == (size[0], size[1], 12, 4)
    ref = jpeg_reader.read_jpeg(file)
    _, want = O.decode(ref.planes, [ref.quanta[c] for c in range(4)], factors, size, precision=12)
    assert rect.shape == (size[1], size[0], 4) and (rect.reshape(-1) == np.asarray(want).reshape(-1)).all()
    chained = J.Rectangular.decompress(ctx, data.tobytes()).host_values()
    assert (rect.reshape(-1) == chained.reshape(-1)).all()
    # cosited, and only the first three components recognised (the fourth takes part in the scale only)
    _, rect3 = J.Rectangular.decompress_to_host(ctx, data.tobytes(), cosite=True, recognized=3)
    _, want3 = O.decode(ref.planes[:3], [ref.quanta[c] for c in range(3)], factors[:3], size, precision=12, cosited=True,
                        scale=(max(f[0] for f in factors), max(f[1] for f in factors)))
    assert rect3.shape == (size[1], size[0], 3) and (rect3.reshape(-1) == np.asarray(want3).reshape(-1)).all()
    # an 8-bit file of the reference through the same entry point: Rectangular of ycc8 == the .ycc gold's samples
    with pytest.raises(J.JpegAmdError):
        J.Rectangular.decompress_to_host(ctx, data.tobytes(), recognized=5)
