"""GPU parity at BASELINE.json's full sizes (configs 3, 4 and a slice of config 5).

The threaded oracle makes a direct comparison affordable on the GPU box's host (256 threads);
size-independent properties are checked as well: a batch decodes to exactly what its images
decode to one by one (independent images), and an image decodes to the same
pixels wherever it sits in a batch (stride handling)."""
import ctypes as C
import os

import numpy as np
import pytest

from oracle import oracle as O

pytestmark = pytest.mark.gpu
THREADS = min(64, os.cpu_count() or 1)
FACTORS = [(2, 2), (1, 1), (1, 1)]


@pytest.fixture(scope="module")
def env():
    import torch
    import jpeg_amd as J
    from jpeg_amd import _lib, synth
    ctx = J.Context(0)
    q = np.stack([J.compression_quanta("luminance", 1.0), J.compression_quanta("chrominance", 1.0)])
    d_q = torch.from_numpy(q.view(np.int16).copy()).to(ctx.torch_device)
    layout = J.Layout("ycc8", {1: J.Component((2, 2), 0), 2: J.Component((1, 1), 1), 3: J.Component((1, 1), 1)})
    return dict(torch=torch, J=J, lib=_lib.lib(), _lib=_lib, synth=synth, ctx=ctx, q=q, d_q=d_q, layout=layout)


def _decode_batch(e, size, planes, n):
    torch, _lib = e["torch"], e["_lib"]
    units = e["layout"].units(size)
    L = e["layout"].c_layout(size, units, [0, 1, 1])
    out = torch.empty((n, size[0] * size[1] * 3), dtype=torch.uint8, device=e["ctx"].torch_device)
    st = e["lib"].jpeg_amd_decode_batch(e["ctx"].handle, C.byref(L), n, _lib.ptr_array([p.data_ptr() for p in planes]),
                                       _lib.size_array([64 * a * b for a, b in units]), e["d_q"].data_ptr(), 0, 2, 0,
                                       _lib.COLOR_RGB8, out.data_ptr(), size[0] * size[1] * 3)
    assert st == 0
    return out


def _oracle_rgb(e, planes_np, size):
    _, rect = O.decode(planes_np, [e["q"][0], e["q"][1], e["q"][1]], FACTORS, size, threads=THREADS)
    return O.unpack_rgb8(rect, 3, threads=THREADS)


def test_config3_one_8192x8192_image(env):
    e = env
    size = (8192, 8192)
    planes = e["synth"].natural_planes_torch(e["layout"].units(size), 1, e["ctx"].torch_device, 31)
    got = _decode_batch(e, size, planes, 1)[0].cpu().numpy().reshape(-1, 3)
    want = _oracle_rgb(e, [p[0].cpu().numpy() for p in planes], size)
    assert (got == want).all(), f"{(got != want).sum()} of {got.size} bytes differ"


def test_config4_4096x4096_encode(env):
    e = env
    torch, _lib = e["torch"], e["_lib"]
    size = (4096, 4096)
    px = e["synth"].smooth_rgb_torch(size[0], size[1], 1, e["ctx"].torch_device)
    units = e["layout"].units(size)
    L = e["layout"].c_layout(size, units, [0, 1, 1])
    coefs = [torch.empty(64 * a * b, dtype=torch.int16, device=e["ctx"].torch_device) for a, b in units]
    st = e["lib"].jpeg_amd_encode_batch(e["ctx"].handle, C.byref(L), 1, px.data_ptr(), 0, _lib.COLOR_RGB8, e["d_q"].data_ptr(),
                                       0, 2, _lib.ptr_array([c.data_ptr() for c in coefs]), _lib.size_array([0, 0, 0]))
    assert st == 0
    want = O.encode(px[0].cpu().numpy(), size, FACTORS, [e["q"][0], e["q"][1], e["q"][1]], threads=THREADS)
    for c, w in zip(coefs, want):
        assert (c.cpu().numpy().reshape(w.shape) == w).all()


@pytest.mark.parametrize("size,n,ycc", [((8192, 2176), 1, False), ((640, 472), 52, False), ((1920, 1080), 9, True)], ids=lambda v: str(v))
def test_420_encode_of_several_rounds_integer_box_filter(env, size, n, ycc):
    """Launches of more workgroups than are resident at once take the 4:2:0 encode kernel whose 2 x 2 box filter runs in the
    integer domain (k_encode_fused<..., POOLI = true>, round 4; one-round launches like config 4 keep the float form): one large
    frame and two batches (RGB and YCbCr input, ragged sizes), every coefficient against the oracle."""
    e = env
    torch, _lib = e["torch"], e["_lib"]
    w, h = size
    dev = e["ctx"].torch_device
    g = torch.Generator(device=dev); g.manual_seed(w + h + n)
    px = torch.randint(0, 256, (n, h * w * 3), dtype=torch.uint8, device=dev, generator=g)
    px[:, : (h * w * 3) // 2] //= 3     # a dark half: the box filter sees small and large sums
    units = e["layout"].units(size)
    L = e["layout"].c_layout(size, units, [0, 1, 1])
    coefs = [torch.empty((n, 64 * a * b), dtype=torch.int16, device=dev) for a, b in units]
    st = e["lib"].jpeg_amd_encode_batch(e["ctx"].handle, C.byref(L), n, px.data_ptr(), h * w * 3, _lib.COLOR_YCC8 if ycc else _lib.COLOR_RGB8,
                                       e["d_q"].data_ptr(), 0, 2, _lib.ptr_array([c.data_ptr() for c in coefs]),
                                       _lib.size_array([64 * a * b for a, b in units]))
    assert st == 0
    for i in sorted({0, n // 2, n - 1}):
        pix = px[i].cpu().numpy().reshape(-1, 3)
        if ycc:
            planar = O.decompose(O.pack_ycc8(pix, 3).reshape(h, w, 3), size, FACTORS, (2, 2))
            want = [O.fdct_plane(p, q) for p, q in zip(planar, [e["q"][0], e["q"][1], e["q"][1]])]
        else:
            want = O.encode(pix, size, FACTORS, [e["q"][0], e["q"][1], e["q"][1]], threads=THREADS)
        for c, wnt in zip(coefs, want):
            assert (c[i].cpu().numpy().reshape(wnt.shape) == wnt).all(), f"image {i}"


@pytest.mark.parametrize("size,n", [((2048, 64), 1), ((2048, 128), 1), ((4096, 192), 1), ((2048, 1024), 2), ((2048, 128), 5), ((6144, 64), 3),
                                    ((512, 256), 9), ((256, 64), 40), ((1280, 1016), 2), ((768, 1000), 3),
                                    ((2048, 96), 2), ((512, 112), 7), ((3840, 2160), 1), ((256, 32), 9), ((768, 480), 3)],
                         ids=lambda v: str(v))
def test_quad_shaped_images_match_oracle(env, size, n):
    """k_quad420's stack walk (one launch, the waves of a workgroup share a chroma tile): images one stack high (no
    neighbour above or below), two, three (both kinds of neighbour), the plane's left and right edge in every row of stacks,
    batches, and shapes with a short last stack (the last five) -- every pixel against the oracle."""
    e = env
    planes = e["synth"].natural_planes_torch(e["layout"].units(size), n, e["ctx"].torch_device, 400 + size[1] + n)
    batch = _decode_batch(e, size, planes, n)
    for i in range(n):
        want = _oracle_rgb(e, [p[i].cpu().numpy() for p in planes], size)
        got = batch[i].cpu().numpy().reshape(-1, 3)
        assert (got == want).all(), f"image {i}: {(got != want).sum()} bytes differ"


def test_config5_slice_batch_of_1080p(env):
    """64 of config 5's 1920x1080 images (MCU grid 120 x 68, last MCU row half padded)."""
    e = env
    size, n = (1920, 1080), 64
    planes = e["synth"].natural_planes_torch(e["layout"].units(size), n, e["ctx"].torch_device, 77)
    batch = _decode_batch(e, size, planes, n)
    # a few images against the oracle
    for i in (0, 17, n - 1):
        want = _oracle_rgb(e, [p[i].cpu().numpy() for p in planes], size)
        assert (batch[i].cpu().numpy().reshape(-1, 3) == want).all(), f"image {i}"
    # batch == images one by one (independent images; strides)
    for i in (1, 40):
        single = _decode_batch(e, size, [p[i:i + 1] for p in planes], 1)[0]
        assert e["torch"].equal(single, batch[i])
    # cheap whole-batch witness: every image differs from its neighbour (no aliasing of strides)
    sums = batch.to(e["torch"].int64).sum(dim=1).cpu().numpy()
    assert len(set(sums.tolist())) == n


def test_config5_all_4096_images_of_1080p_on_one_gpu(env):
    """BASELINE.json configs[4] in its stated size on ONE GPU (the N = 1 point of the sharded job: 25.5 GB of coefficients,
    25.5 GB of pixels): 4096 independent 1920x1080 images in one call -- the ticket walk over 524 288 stacks, the mixed
    column cut, 32-bit stack indices.  Eight images spread over the batch against the oracle, the same eight decoded again
    one by one (a batch is its images), and a whole-batch witness: every image differs from every other one.
    A GPU with less than 64 GB free runs the largest power of two that fits (>= 1024 images)."""
    e = env
    torch = e["torch"]
    size, n = (1920, 1080), 4096
    free, _ = torch.cuda.mem_get_info(e["ctx"].torch_device)
    per_image = 2 * sum(64 * a * b for a, b in e["layout"].units(size)) + size[0] * size[1] * 3
    while n > 1024 and n * per_image + (8 << 30) > free:
        n //= 2
    assert n * per_image + (4 << 30) <= free, "not enough device memory for 1024 images of 1080p"
    planes = e["synth"].natural_planes_torch(e["layout"].units(size), n, e["ctx"].torch_device, 4096)
    batch = _decode_batch(e, size, planes, n)
    spread = [0, 1, n // 8 + 3, n // 2 - 1, n // 2, (5 * n) // 8 + 17, n - 2, n - 1]
    for i in spread:
        want = _oracle_rgb(e, [p[i].cpu().numpy() for p in planes], size)
        assert (batch[i].cpu().numpy().reshape(-1, 3) == want).all(), f"image {i} of {n}"
        single = _decode_batch(e, size, [p[i:i + 1] for p in planes], 1)[0]
        assert torch.equal(single, batch[i]), f"image {i}: batch != single"
    # whole-batch witness, in chunks (an int64 copy of the batch would not fit): a position-weighted checksum per image
    # (plain byte sums of 4096 images collide by chance)
    weights = (torch.arange(size[0] * size[1] * 3, device=batch.device, dtype=torch.int64) % 65521) + 1
    sums = torch.cat([(batch[i:i + 64].to(torch.int64) * weights).sum(dim=1) for i in range(0, n, 64)]).cpu().numpy()
    assert len(set(sums.tolist())) == n
    del batch, planes
    torch.cuda.empty_cache()


@pytest.mark.parametrize("size", [(65535, 24), (24, 65535)], ids=["widest", "tallest"])
def test_extreme_aspect_ratios_at_the_frame_header_limit(size):
    """The frame header stores width and height in 16 bits (decode.swift:793-800): the widest
    and the tallest image a JPEG can describe, 4:2:0, through the fused decode and encode."""
    import jpeg_amd as J
    from oracle import oracle as O
    ctx = J.Context()
    rng = np.random.default_rng(7)
    layout = J.Layout("ycc8", {1: J.Component((2, 2), 0), 2: J.Component((1, 1), 1), 3: J.Component((1, 1), 1)})
    units = layout.units(size)
    planes = []
    for ux, uy in units:
        c = rng.integers(-300, 300, (uy, ux, 64)).astype(np.int16)
        c[..., 6:] //= 16
        planes.append(c)
    quanta = [J.compression_quanta("luminance", 1.0), J.compression_quanta("chrominance", 1.0)]
    spectral = J.Spectral.from_host(ctx, size, layout, planes, quanta, q=[0, 1, 1])
    got = spectral.decode(J.RGB).cpu().numpy()
    _, rect = O.decode(planes, [quanta[0], quanta[1], quanta[1]], [(2, 2), (1, 1), (1, 1)], size, threads=8)
    want = O.unpack_rgb8(rect, 3, threads=8)
    assert (got == want).all()
    coef = J.Rectangular.encode(ctx, size, layout, want, {0: quanta[0], 1: quanta[1]}, J.RGB).host_planes()
    ref = O.encode(want, size, [(2, 2), (1, 1), (1, 1)], [quanta[0], quanta[1], quanta[1]], threads=8)
    for a, b in zip(coef, ref):
        assert (a == b).all()


def test_one_image_in_bands_across_ranks_is_bit_identical():
    """jpeg_amd.dist.band on the device path: a 2048 x 1544 4:2:0 image decoded as 3 bands (each
    with its one-MCU-row halo, zero-copy row slices of the resident planes) equals the whole-image
    fused decode -- the strong-scaling split of SURVEY.md 8e, ranks simulated one after another."""
    import torch
    import jpeg_amd as J
    from jpeg_amd import dist as jd, synth
    ctx = J.Context()
    size = (2048, 1544)
    layout = J.Layout("ycc8", {1: J.Component((2, 2), 0), 2: J.Component((1, 1), 1), 3: J.Component((1, 1), 1)})
    units = layout.units(size)
    planes = [p[0] for p in synth.natural_planes_torch(units, 1, ctx.torch_device, 11)]
    quanta = [J.compression_quanta("luminance", 1.0), J.compression_quanta("chrominance", 1.0)]
    whole = J.Spectral(ctx, size, layout, [p.view(uy, ux, 64) for p, (ux, uy) in zip(planes, units)], quanta, [0, 1, 1]).decode(J.RGB)
    whole = whole.view(size[1], size[0], 3)
    world, rows = 3, []
    for rank in range(world):
        plan = jd.band(size, layout.scale, rank, world)
        sub = []
        for p, c, (ux, uy) in zip(planes, layout.planes, units):
            u0, u1 = jd.band_units(plan, c.factor[1], uy)
            sub.append(p.view(uy, ux, 64)[u0:u1])
        px = J.Spectral(ctx, (size[0], plan["height"]), layout, sub, quanta, [0, 1, 1]).decode(J.RGB).view(plan["height"], size[0], 3)
        y0, y1 = plan["rows"]
        rows.append(px[plan["skip"]:plan["skip"] + (y1 - y0)])
    assert torch.equal(torch.cat(rows), whole)


def test_planes_beyond_4GiB_are_addressed_in_64_bits(env):
    """A 65520 x 33024 4:2:0 image: luma coefficient plane 4.3 GB, RGB 6.5 GB -- byte offsets no
    longer fit 32 bits (the frame header allows 65535 x 65535).  No oracle run at this size: bands
    of whole MCU rows decoded / encoded as small sub-images through the SAME (oracle-checked)
    kernels, from row slices of the resident buffers, must equal the whole-image result."""
    e = env
    torch, _lib, lib, ctx = e["torch"], e["_lib"], e["lib"], e["ctx"]
    dev = ctx.torch_device
    W, H = 65520, 33024
    units = e["layout"].units((W, H))
    assert 128 * units[0][0] * units[0][1] > 1 << 32 and 3 * W * H > 1 << 32
    g = torch.Generator(device=dev); g.manual_seed(5)
    planes = []
    for ux, uy in units:
        p = torch.randint(-40, 41, (uy * ux, 64), dtype=torch.int16, device=dev, generator=g)
        p[:, 10:] = 0
        p[:, 0] *= 8
        planes.append(p.view(-1))
    n_mcu = H // 16

    def decode(m0, m1, out):
        size = (W, (m1 - m0) * 16)
        L = e["layout"].c_layout(size, e["layout"].units(size), [0, 1, 1])
        ptrs = [p.data_ptr() + 2 * 64 * ux * (m0 * f) for p, (ux, _), f in zip(planes, units, (2, 1, 1))]
        st = lib.jpeg_amd_decode_batch(ctx.handle, C.byref(L), 1, _lib.ptr_array(ptrs), _lib.size_array([0, 0, 0]),
                                       e["d_q"].data_ptr(), 0, 2, 0, _lib.COLOR_RGB8, out.data_ptr(), 0)
        assert st == 0

    rgb = torch.empty(3 * W * H, dtype=torch.uint8, device=dev)
    decode(0, n_mcu, rgb)
    rows = rgb.view(H, 3 * W)
    straddle = (1 << 32) // (3 * W) // 16          # the MCU row whose pixels cross the 4 GiB offset
    for m0, m1, top, bottom in [(0, 6, 0, 1), (straddle - 4, straddle + 4, 1, 1), (n_mcu - 6, n_mcu, 1, 0)]:
        band = torch.empty(3 * W * (m1 - m0) * 16, dtype=torch.uint8, device=dev)
        decode(m0, m1, band)
        band = band.view(-1, 3 * W)
        # the sub-image replicates chroma at its own top / bottom edge: skip the halo MCU row there
        a, b = 16 * top, band.shape[0] - 16 * bottom
        assert torch.equal(band[a:b], rows[16 * m0 + a:16 * m0 + b]), (m0, m1)

    def encode(m0, m1, coefs):
        size = (W, (m1 - m0) * 16)
        L = e["layout"].c_layout(size, e["layout"].units(size), [0, 1, 1])
        st = lib.jpeg_amd_encode_batch(ctx.handle, C.byref(L), 1, rgb.data_ptr() + 3 * W * 16 * m0, 0, _lib.COLOR_RGB8,
                                       e["d_q"].data_ptr(), 0, 2, _lib.ptr_array([c.data_ptr() for c in coefs]),
                                       _lib.size_array([0, 0, 0]))
        assert st == 0

    del planes
    whole = [torch.empty(64 * ux * uy, dtype=torch.int16, device=dev) for ux, uy in units]
    encode(0, n_mcu, whole)
    last_luma = (1 << 32) // (128 * units[0][0]) // 2   # the MCU row whose luma blocks cross 4 GiB
    for m0, m1 in [(0, 4), (last_luma - 2, last_luma + 2), (n_mcu - 4, n_mcu)]:
        part = [torch.empty(64 * ux * (m1 - m0) * f, dtype=torch.int16, device=dev) for (ux, _), f in zip(units, (2, 1, 1))]
        encode(m0, m1, part)
        for c, w, (ux, _), f in zip(part, whole, units, (2, 1, 1)):
            assert torch.equal(c, w[64 * ux * m0 * f:64 * ux * m1 * f]), (m0, m1)


def test_single_image_and_batch_of_420_agree(env):
    """One 2048 x 1024 4:2:0 image and a batch of the same geometry (the stack numbering runs through the batch,
    k_quad420): same pixels, both equal to the oracle."""
    e = env
    size, n = (2048, 1024), 3
    planes = e["synth"].natural_planes_torch(e["layout"].units(size), n, e["ctx"].torch_device, 5)
    batch = _decode_batch(e, size, planes, n)
    for i in range(n):
        single = _decode_batch(e, size, [p[i:i + 1] for p in planes], 1)[0]
        assert e["torch"].equal(single, batch[i]), f"image {i}: the two 4:2:0 paths differ"
    want = _oracle_rgb(e, [p[1].cpu().numpy() for p in planes], size)
    assert (batch[1].cpu().numpy().reshape(-1, 3) == want).all()


def test_440_batch_in_pairs_of_strips_equals_single_images_in_single_strips(env):
    """4:4:0 (k_luma_fused): a batch with more strips than waves are resident walks PAIRS of strip rows that share a chroma tile, a small
    call single strips (LumaArgs::pair, decided per call).  40 images of 600 x 437 -- 5 x 14 strips each, an odd number of strip rows, so
    every image ends in a half pair -- decoded in one call (2 800 strips: pairs) and one by one (70: single): identical, and equal
    to the oracle."""
    e = env
    torch, _lib, J = e["torch"], e["_lib"], e["J"]
    layout = J.Layout("ycc8", {1: J.Component((1, 2), 0), 2: J.Component((1, 1), 1), 3: J.Component((1, 1), 1)})
    size, n = (600, 437), 40
    units = layout.units(size)
    L = layout.c_layout(size, units, [0, 1, 1])
    planes = e["synth"].natural_planes_torch(units, n, e["ctx"].torch_device, 44)
    npx = size[0] * size[1] * 3

    def decode(ps, k):
        out = torch.empty((k, npx), dtype=torch.uint8, device=e["ctx"].torch_device)
        st = e["lib"].jpeg_amd_decode_batch(e["ctx"].handle, C.byref(L), k, _lib.ptr_array([p.data_ptr() for p in ps]),
                                           _lib.size_array([64 * a * b for a, b in units]), e["d_q"].data_ptr(), 0, 2, 0,
                                           _lib.COLOR_RGB8, out.data_ptr(), npx)
        assert st == 0
        return out

    batch = decode(planes, n)
    for i in range(n):
        assert torch.equal(decode([p[i:i + 1] for p in planes], 1)[0], batch[i]), i
    _, rect = O.decode([p[7].cpu().numpy() for p in planes], [e["q"][0], e["q"][1], e["q"][1]], [(1, 2), (1, 1), (1, 1)], size, threads=THREADS)
    assert (batch[7].cpu().numpy().reshape(-1, 3) == O.unpack_rgb8(rect, 3, threads=THREADS)).all()
