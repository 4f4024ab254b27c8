"""Every shape of 4:2:0 image goes through ONE kernel, k_quad420 (kernels_quad.hip): stacks of four 32 x 2-block strips or of
two 16 x 4-block strips, whichever covers the plane with fewer strips.  Shapes that exercise each of its edge mechanisms --
a partial last tile column (clamped fetches, the plane edge repaired inside the tile), a partial last strip, a SHORT last
stack (strips of the stack that lie wholly below the image), a plane that ends in the middle of a wave's window, odd sizes
(the byte-wise store tail), one-block images -- against the oracle, both colour targets, plus the encode mirror."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

SIZES = [
    (2048, 1024),    # whole stacks, whole columns (the shape of config 3)
    (2048, 1540),    # 96.25 strips of 32 x 2: a short last stack and a partial last strip
    (1920, 1080),    # config 5: one image takes 16 x 4 strips (15 x 34, 135 stacks); long batches the mixed cut (7 columns of 32 x 2 + 1 of 16 x 4)
    (3840, 2160),    # 135 strip rows of 32 x 2 (33.75 stacks) -> 16 x 4 strips, 30 x 68
    (2064, 192),     # partial last column of 32 x 2 strips
    (4112, 520),     # partial last column and partial last strip
    (1000, 700), (520, 24), (17, 17), (8, 8), (16, 16), (24, 40),
    (640, 320), (128, 64), (128, 48),   # 16 x 4 strips: whole stacks, one stack, a plane that ends inside the first strip
    (256, 112), (256, 144),             # 32 x 2 strips: short stacks of 3 / a full stack + one strip
    (304, 304),
]


@pytest.fixture(scope="module")
def env():
    import jpeg_amd as J
    from oracle import oracle as O
    return J, O, J.Context(0)


@pytest.mark.parametrize("size", SIZES, ids=lambda s: "%dx%d" % s)
def test_420_decode_and_encode_any_shape(env, size):
    J, O, ctx = env
    rng = np.random.default_rng(11 + size[0] * 7 + size[1])
    quanta = [rng.integers(1, 30, 64).astype(np.uint16) for _ in range(2)]
    layout = J.Layout("ycc8", {1: J.Component((2, 2), 0), 2: J.Component((1, 1), 1), 3: J.Component((1, 1), 1)})
    units = layout.units(size)
    planes = [np.clip(rng.laplace(0, 30, (uy, ux, 64)), -1000, 1000).astype(np.int16) for ux, uy in units]
    spectral = J.Spectral.from_host(ctx, size, layout, planes, quanta, q=[0, 1, 1])
    _, rect = O.decode(planes, [quanta[0], quanta[1], quanta[1]], [(2, 2), (1, 1), (1, 1)], size, threads=8)
    rgb = O.unpack_rgb8(rect, 3, threads=8)
    assert (spectral.decode(J.RGB).cpu().numpy() == rgb).all()
    assert (spectral.decode(J.YCbCr).cpu().numpy() == O.unpack_ycc8(rect, 3)).all()
    coef = J.Rectangular.encode(ctx, size, layout, rgb, {0: quanta[0], 1: quanta[1]}, J.RGB).host_planes()
    want = O.encode(rgb, size, [(2, 2), (1, 1), (1, 1)], [quanta[0], quanta[1], quanta[1]], threads=8)
    assert all((a == b).all() for a, b in zip(coef, want))


@pytest.mark.parametrize("size,n", [((1920, 1080), 5), ((256, 112), 7), ((2064, 192), 3), ((128, 48), 9)])
def test_420_batches_of_edge_shapes(env, size, n):
    """Stacks are numbered through the whole batch: images whose last stack is short or whose stack count is odd (two
    stacks per workgroup with 16 x 4 strips) must not leak into their neighbours."""
    J, O, ctx = env
    import torch
    from jpeg_amd import _lib
    import ctypes as C
    rng = np.random.default_rng(5)
    quanta = [rng.integers(1, 30, 64).astype(np.uint16) for _ in range(2)]
    layout = J.Layout("ycc8", {1: J.Component((2, 2), 0), 2: J.Component((1, 1), 1), 3: J.Component((1, 1), 1)})
    units = layout.units(size)
    W, H = size
    batch = [[np.clip(rng.laplace(0, 40, (uy, ux, 64)), -1000, 1000).astype(np.int16) for ux, uy in units] for _ in range(n)]
    dev = ctx.torch_device
    d_planes = [torch.from_numpy(np.stack([b[p] for b in batch])).to(dev) for p in range(3)]
    d_q = torch.from_numpy(np.stack(quanta).view(np.int16)).to(dev)
    out = torch.zeros((n, W * H * 3), dtype=torch.uint8, device=dev)
    L = layout.c_layout(size, units, [0, 1, 1])
    strides = _lib.size_array([64 * a * b for a, b in units])
    st = _lib.lib().jpeg_amd_decode_batch(ctx.handle, C.byref(L), n, _lib.ptr_array([p.data_ptr() for p in d_planes]), strides,
                                          d_q.data_ptr(), 0, 2, 0, _lib.COLOR_RGB8, out.data_ptr(), W * H * 3)
    assert st == 0
    got = out.cpu().numpy()
    for i in range(n):
        _, rect = O.decode(batch[i], [quanta[0], quanta[1], quanta[1]], [(2, 2), (1, 1), (1, 1)], size, threads=8)
        assert (got[i] == O.unpack_rgb8(rect, 3, threads=8).reshape(-1)).all(), i


@pytest.mark.parametrize("size,n", [((384, 272), 1400), ((1920, 1080), 93)])
def test_420_large_batches_take_the_mixed_column_cut(env, size, n):
    """Batches that are many trips long are cut into 32 x 2 strips for the whole columns plus ONE column of 16 x 4 strips
    for a remainder of at most 16 blocks (two launches, quad_cut in kernels_quad.hip): 384 x 272 is 1 + 1 columns (8 stacks
    per image instead of 9), 1920 x 1080 7 + 1 (128 instead of 135).  The seam between the two launches runs through
    the middle of the image: a sample of the batch against the oracle."""
    J, O, ctx = env
    import torch
    from jpeg_amd import _lib
    import ctypes as C
    rng = np.random.default_rng(17)
    quanta = [rng.integers(1, 30, 64).astype(np.uint16) for _ in range(2)]
    layout = J.Layout("ycc8", {1: J.Component((2, 2), 0), 2: J.Component((1, 1), 1), 3: J.Component((1, 1), 1)})
    units = layout.units(size)
    W, H = size
    dev = ctx.torch_device
    pool = [np.clip(rng.laplace(0, 40, (8, uy, ux, 64)), -1000, 1000).astype(np.int16) for ux, uy in units]   # 8 distinct images
    host = [p[np.arange(n) % 8] for p in pool]
    d_planes = [torch.from_numpy(h).to(dev) for h in host]
    d_q = torch.from_numpy(np.stack(quanta).view(np.int16)).to(dev)
    out = torch.zeros((n, W * H * 3), dtype=torch.uint8, device=dev)
    L = layout.c_layout(size, units, [0, 1, 1])
    strides = _lib.size_array([64 * a * b for a, b in units])
    st = _lib.lib().jpeg_amd_decode_batch(ctx.handle, C.byref(L), n, _lib.ptr_array([p.data_ptr() for p in d_planes]), strides,
                                          d_q.data_ptr(), 0, 2, 0, _lib.COLOR_RGB8, out.data_ptr(), W * H * 3)
    assert st == 0
    for i in sorted({0, 1, n // 3, n // 2, n - 2, n - 1}):
        _, rect = O.decode([h[i] for h in host], [quanta[0], quanta[1], quanta[1]], [(2, 2), (1, 1), (1, 1)], size, threads=8)
        assert (out[i].cpu().numpy() == O.unpack_rgb8(rect, 3, threads=8).reshape(-1)).all(), i


@pytest.mark.parametrize("size,n", [((1920, 1080), 128), ((256, 64), 13000), ((272, 200), 2100)])
def test_420_long_calls_take_the_ticket_walk(env, size, n):
    """Calls of 16 or more trips per workgroup (>= 12 288 stacks in a launch) hand their stacks out through a ticket counter
    instead of the static stride (k_quad420, dynamic walk): every image of the batch against the oracle's result for its
    source (a pool of 8 distinct images), twice in a row on the same context (the counter is zeroed per launch)."""
    J, O, ctx = env
    import torch
    from jpeg_amd import _lib
    import ctypes as C
    rng = np.random.default_rng(23)
    quanta = [rng.integers(1, 30, 64).astype(np.uint16) for _ in range(2)]
    layout = J.Layout("ycc8", {1: J.Component((2, 2), 0), 2: J.Component((1, 1), 1), 3: J.Component((1, 1), 1)})
    units = layout.units(size)
    W, H = size
    dev = ctx.torch_device
    pool = [np.clip(rng.laplace(0, 40, (8, uy, ux, 64)), -1000, 1000).astype(np.int16) for ux, uy in units]
    idx = rng.integers(0, 8, n)
    d_planes = [torch.from_numpy(p[idx]).to(dev) for p in pool]
    d_q = torch.from_numpy(np.stack(quanta).view(np.int16)).to(dev)
    L = layout.c_layout(size, units, [0, 1, 1])
    strides = _lib.size_array([64 * a * b for a, b in units])
    want = []
    for k in range(8):
        _, rect = O.decode([p[k] for p in pool], [quanta[0], quanta[1], quanta[1]], [(2, 2), (1, 1), (1, 1)], size, threads=8)
        want.append(torch.from_numpy(O.unpack_rgb8(rect, 3, threads=8).reshape(-1)).to(dev))
    want = torch.stack(want)[torch.from_numpy(idx).to(dev)]
    for rep in range(2):
        out = torch.zeros((n, W * H * 3), dtype=torch.uint8, device=dev)
        st = _lib.lib().jpeg_amd_decode_batch(ctx.handle, C.byref(L), n, _lib.ptr_array([p.data_ptr() for p in d_planes]), strides,
                                              d_q.data_ptr(), 0, 2, 0, _lib.COLOR_RGB8, out.data_ptr(), W * H * 3)
        assert st == 0
        bad = (out != want).any(dim=1)
        assert not bool(bad.any()), (rep, int(bad.sum()), int(torch.nonzero(bad)[0]))
