"""The colour stage of the fused decode kernels on the device, every input: the kernels convert with v_cvt_pk_u8_f32 under
round-toward-zero (trunc_pack*, fused_common.hpp) after one FMA for R and B and two for G.  tests/test_colour_rounding.py
proves that arithmetic on the CPU for all 2^24 (Y, Cb, Cr); this test pushes all 2^24 through the 4:4:4 kernel itself and
a sample through the 4:2:0 kernel (constant chroma planes, every Y), against the reference's formula (jpeg.swift:441-453:
x = (y + m_cb (cb - 128)) + m_cr (cr - 128), clamp, truncate) evaluated in binary32 with numpy."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
f32 = np.float32


def _rgb(y, cb, cr):
    y, pb, pr = y.astype(f32), cb.astype(f32) - f32(128), cr.astype(f32) - f32(128)
    r = (y + (f32(0.0) * pb).astype(f32)).astype(f32) + (f32(1.40200) * pr).astype(f32)
    g = (y + (f32(-0.34414) * pb).astype(f32)).astype(f32) + (f32(-0.71414) * pr).astype(f32)
    b = (y + (f32(1.77200) * pb).astype(f32)).astype(f32) + (f32(0.0) * pr).astype(f32)
    return np.stack([np.clip(c.astype(f32), f32(0), f32(255)).astype(np.uint8) for c in (r, g, b)], axis=-1)


def _dc_plane(values):
    """Coefficient plane [uy][ux][64] whose blocks decode (all-ones table) to the constant sample `values[uy][ux]`:
    sample = floor(128.5 + DC / 8)."""
    p = np.zeros(values.shape + (64,), np.int16)
    p[..., 0] = 8 * (values.astype(np.int32) - 128)
    return p


@pytest.fixture(scope="module")
def env():
    import jpeg_amd as J
    return J, J.Context(0)


def test_every_y_cb_cr_through_the_444_kernel(env):
    J, ctx = env
    ones = np.ones(64, np.uint16)
    layout = J.Layout("ycc8", {1: J.Component((1, 1), 0), 2: J.Component((1, 1), 1), 3: J.Component((1, 1), 1)})
    size = (2048, 2048)                       # 256 x 256 blocks: block (j, i) carries Y = i, Cr = j
    yy, cr = np.meshgrid(np.arange(256), np.arange(256))
    for cb in range(256):
        planes = [_dc_plane(yy), _dc_plane(np.full((256, 256), cb)), _dc_plane(cr)]
        got = J.Spectral.from_host(ctx, size, layout, planes, [ones, ones], q=[0, 1, 1]).decode(J.RGB).cpu().numpy().reshape(2048, 2048, 3)
        want = _rgb(yy, np.full((256, 256), cb), cr)                     # per block
        assert (got.reshape(256, 8, 256, 8, 3) == want[:, None, :, None, :]).all(), cb


def test_every_y_against_sampled_chroma_through_the_420_kernel(env):
    J, ctx = env
    ones = np.ones(64, np.uint16)
    layout = J.Layout("ycc8", {1: J.Component((2, 2), 0), 2: J.Component((1, 1), 1), 3: J.Component((1, 1), 1)})
    size = (512, 64)                          # 64 x 8 luma blocks, Y = 4 * (i % 64) + (j % 4) ... every value 0 ... 255 twice
    yy = (np.arange(8)[:, None] * 64 + np.arange(64)[None, :]) % 256
    rng = np.random.default_rng(2)
    pairs = [(0, 0), (255, 255), (0, 255), (255, 0), (128, 128), (127, 129)] + [tuple(p) for p in rng.integers(0, 256, (58, 2))]
    for cb, cr in pairs:
        planes = [_dc_plane(yy), _dc_plane(np.full((4, 32), cb)), _dc_plane(np.full((4, 32), cr))]
        got = J.Spectral.from_host(ctx, size, layout, planes, [ones, ones], q=[0, 1, 1]).decode(J.RGB).cpu().numpy().reshape(64, 512, 3)
        want = _rgb(yy, np.full((8, 64), cb), np.full((8, 64), cr))
        assert (got.reshape(8, 8, 64, 8, 3) == want[:, None, :, None, :]).all(), (cb, cr)
