"""Exhaustive CPU proof of the colour stage's shortcuts in the fused decode kernels (kernels_quad.hip, kernels_fused.hip).

The reference clamps the colour-matrix result to [0, 255] and TRUNCATES (jpeg.swift:343-354, 441-453).  The kernels pack with
v_cvt_pk_u8_f32 under round-toward-zero (trunc_pack*, fused_common.hpp; tools/probe_cvt_round.hip shows that the instruction
follows the wave's rounding mode): saturate + truncate, the same conversion for every float.  What remains to prove is the
arithmetic in front of it: the kernels use one FMA for R and B and two for G where the reference rounds every product and
every sum.  These tests check EVERY input combination in binary32.
"""
import numpy as np

f32 = np.float32


def _ref(x):   # clamp then truncate toward zero
    return np.clip(x, f32(0), f32(255)).astype(np.int32)


def _hw(z):    # v_cvt_pk_u8_f32 under round-toward-zero: truncate, saturate
    return np.clip(np.trunc(z), 0, 255).astype(np.int32)


def test_the_truncating_convert_is_the_references_conversion():
    x = np.concatenate([np.linspace(-300, 600, 90001, dtype=f32), np.arange(-2, 258, dtype=f32), np.nextafter(np.arange(0, 257, dtype=f32), f32(-1e9))])
    assert np.array_equal(_ref(x), _hw(x))


def test_red_and_blue_every_input():
    y = np.arange(256, dtype=f32)[:, None]
    c = np.arange(-128, 128, dtype=f32)[None, :]
    for m in (f32(1.40200), f32(1.77200)):
        p = (m * c).astype(f32)                       # one rounding, as in the reference
        want = _ref((y + p).astype(f32))
        # the kernel: fma(m, c, y), one rounding of the exact m*c + y.  float64 holds that sum exactly (24-bit x 8-bit
        # product, 8-bit addend)
        exact = np.float64(m) * c.astype(np.float64) + y.astype(np.float64)
        assert np.all(exact - y.astype(np.float64) == np.float64(m) * c.astype(np.float64))   # no f64 rounding
        assert np.array_equal(want, _hw(exact.astype(f32)))


def _fma(p, q, r):
    """binary32 fma of a constant p, small integers q and a binary32 r: the float64 value
    p*q + r is exact here (< 40 significant bits), so one cast rounds once, like the hardware."""
    return (np.float64(p) * q.astype(np.float64) + r.astype(np.float64)).astype(f32)


def test_green_two_fused_multiply_adds_every_input():
    y = np.arange(256, dtype=f32)[:, None, None]
    pb = np.arange(-128, 128, dtype=f32)[None, :, None]
    pr = np.arange(-128, 128, dtype=f32)[None, None, :]
    a, b = f32(-0.34414), f32(-0.71414)
    x = ((y + (a * pb).astype(f32)).astype(f32) + (b * pr).astype(f32)).astype(f32)
    want = _ref(x)
    got = _hw(_fma(b, pr, _fma(a, pb, y + 0 * pb)))
    assert np.array_equal(want, got)                  # the kernel's form: exact on all 2^24 triples
    other = _hw(_fma(a, pb, _fma(b, pr, y + 0 * pr)))
    assert (other != want).any()                      # the other association is not


def test_encode_matrix_only_the_exact_half_products_may_be_fused():
    """k_encode_fused (kernels_encode.hip, rgb_to_ycc) evaluates RGB.ycc (jpeg.swift:463-478) op for op except for the two
    steps whose product is exact -- 0.5 * b in Cb and 0.5 * r in Cr -- which are one FMA each.  All 2^24 (r, g, b):
    those two placements give the reference's integer everywhere, every other FMA placement changes some result."""
    LD = np.longdouble   # 64-bit significand: holds a 24 x 8-bit product plus a 24-bit addend of this range exactly

    def fma(a, x, c):
        return (LD(a) * x.astype(LD) + c.astype(LD)).astype(f32)

    v = np.arange(256, dtype=f32)
    r, g, b = (a.ravel() for a in np.meshgrid(v, v, v, indexing="ij"))
    rows = {"y": (0.0, 0.2990, 0.5870, 0.1140), "cb": (128.0, -0.1687, -0.3313, 0.5), "cr": (128.0, 0.5, -0.4187, -0.0813)}
    exact_placements = {"y": {(0, 0, 0), (1, 0, 0)}, "cb": {(0, 0, 0), (0, 0, 1)}, "cr": {(0, 0, 0), (1, 0, 0)}}
    for name, (m0, mr, mg, mb) in rows.items():
        m0, mr, mg, mb = f32(m0), f32(mr), f32(mg), f32(mb)
        want = np.floor(((m0 + mr * r) + mg * g) + mb * b)
        for s1 in (0, 1):
            for s2 in (0, 1):
                for s3 in (0, 1):
                    x = fma(mr, r, np.full_like(r, m0)) if s1 else m0 + mr * r
                    x = fma(mg, g, x) if s2 else x + mg * g
                    x = fma(mb, b, x) if s3 else x + mb * b
                    same = bool(np.array_equal(np.floor(x), want))
                    assert same == ((s1, s2, s3) in exact_placements[name]), (name, s1, s2, s3)


def test_floor_free_chroma_rounding_every_value():
    """k_luma_fused (4:2:0) rounds the upsampled chroma without a floor (upsample.hpp): bytes enter as 2^15 + p + 1/32,
    the two 3a + b steps are exact, and ONE fma(., 1/16, C) rounds to the integer.  Every byte pair for the two exact
    steps, every v = 9a + 3b + 3c + d (0 ... 4080) and both output modes for the rounding."""
    def fma(a, b, c):   # exact in longdouble for these magnitudes, rounded once to binary32
        return (a.astype(np.longdouble) * np.longdouble(b) + c.astype(np.longdouble)).astype(f32) if np.ndim(b) == 0 else \
               (a.astype(np.longdouble) * b.astype(np.longdouble) + c.astype(np.longdouble)).astype(f32)

    byte = np.arange(256, dtype=np.uint32)
    P = ((np.uint32(0x47000008) | (byte << np.uint32(8))).astype(np.uint32)).view(f32)       # what v_perm_b32 builds
    assert np.array_equal(P.astype(np.float64), 32768.0 + byte + 1.0 / 32)
    pn, pf = np.meshgrid(P, P, indexing="ij")
    bn, bf = np.meshgrid(byte.astype(np.float64), byte.astype(np.float64), indexing="ij")
    H = fma(pn.ravel(), 3.0, pf.ravel())                                                          # horizontal 3a + b
    assert np.array_equal(H.astype(np.float64), 131072.0 + (3 * bn + bf).ravel() + 0.125)
    # vertical 3A + B over every pair of horizontal results would be 1021^2 cases of the same exactness argument; check the
    # extremes and a dense sample, then every resulting v for the rounding itself
    h = np.arange(0, 1021, dtype=np.float64)
    Hs = (131072.0 + h + 0.125).astype(f32)
    assert np.array_equal(Hs.astype(np.float64), 131072.0 + h + 0.125)
    hn, hf = np.meshgrid(Hs[::7], Hs[::5], indexing="ij")
    V = fma(hn.ravel(), 3.0, hf.ravel())
    vn, vf = np.meshgrid(h[::7], h[::5], indexing="ij")
    assert np.array_equal(V.astype(np.float64), 524288.0 + (3 * vn + vf).ravel() + 0.5)
    v = np.arange(0, 4081, dtype=np.float64)
    Vall = (524288.0 + v + 0.5).astype(f32)
    assert np.array_equal(Vall.astype(np.float64), 524288.0 + v + 0.5)
    magic = f32(12582912.0)
    for sub128 in (True, False):
        C = f32(12582912.0 - 32768.0 - (128.0 if sub128 else 0.0))
        t = fma(Vall, f32(1.0 / 16), np.full_like(Vall, C))
        got = (t - magic).astype(np.float64)
        want = np.floor((v.astype(f32) * f32(1.0 / 16) + f32(-127.5 if sub128 else 0.5)).astype(f32)).astype(np.float64)  # the staged form
        assert np.array_equal(got, np.floor(v / 16 + 0.5) - (128 if sub128 else 0))
        assert np.array_equal(got, want)


def test_floor_free_chroma_rounding_one_axis_every_byte_pair():
    """k_luma_fused, 4:2:2 and 4:4:0 (one subsampled axis): bytes enter as 2^15 + p + 1/32, the single 3a + b step is exact and
    ONE fma(., 1/4, C) rounds to floor(v / 4 + 1/2) [- 128] -- the value the staged form's floor(fma(v, 1/4, 1/2 [- 128])) and the
    reference's round-half-away of (3a + b) / 4 give.  Every pair of bytes, both output modes."""
    def fma(a, b, c):
        return (a.astype(np.longdouble) * np.longdouble(b) + c.astype(np.longdouble)).astype(f32)

    byte = np.arange(256, dtype=np.uint32)
    P = ((np.uint32(0x47000008) | (byte << np.uint32(8))).astype(np.uint32)).view(f32)
    pn, pf = np.meshgrid(P, P, indexing="ij")
    bn, bf = np.meshgrid(byte.astype(np.float64), byte.astype(np.float64), indexing="ij")
    v = (3 * bn + bf).ravel()
    V = fma(pn.ravel(), 3.0, pf.ravel())
    assert np.array_equal(V.astype(np.float64), 131072.0 + v + 0.125)
    magic = f32(12582912.0)
    for sub128 in (True, False):
        C = f32(12582912.0 - 32768.0 - (128.0 if sub128 else 0.0))
        got = (fma(V, f32(0.25), np.full_like(V, C)) - magic).astype(np.float64)
        assert np.array_equal(got, np.floor(v / 4 + 0.5) - (128 if sub128 else 0))
        staged = np.floor((v.astype(f32) * f32(0.25) + f32(-127.5 if sub128 else 0.5)).astype(f32)).astype(np.float64)
        assert np.array_equal(got, staged)
        # the reference: u0 * 0.75 + u1 * 0.25, rounded half away from zero (decode.swift:4250-4264), [- 128] at the colour stage
        ref = np.floor(((bn.ravel().astype(f32) * f32(0.75)).astype(f32) + (bf.ravel().astype(f32) * f32(0.25)).astype(f32)).astype(f32).astype(np.float64) + 0.5)
        assert np.array_equal(got, ref - (128 if sub128 else 0))


def test_integer_box_filter_of_the_420_encode():
    """k_encode_fused (4:2:0, round 4): four truncated bytes in one dword, ONE v_dot4_u32_u8 with the weights 64, 64, 64, 64, and
    byte 1 of the result is trunc(Float(sum) / 4) (encode.swift:419-421): every sum, and the truncation of every Cb / Cr the
    colour matrix can produce (they lie in [0.5, 255.5], so truncation toward zero IS the reference's clamp + UInt8 conversion)."""
    s = np.arange(0, 4 * 255 + 1, dtype=np.int64)
    assert (64 * s).max() < 1 << 16
    assert np.array_equal(((64 * s) >> 8) & 0xff, (s.astype(f32) / f32(4)).astype(np.int64))
    r, g, b = np.meshgrid(np.arange(256, dtype=f32), np.arange(256, dtype=f32), np.arange(256, dtype=f32), indexing="ij")
    def fma(a, x, c):
        return (np.longdouble(a) * x.astype(np.longdouble) + c.astype(np.longdouble)).astype(f32)
    cb = fma(f32(0.5), b, ((f32(128.0) + f32(-0.1687) * r).astype(f32) + (f32(-0.3313) * g).astype(f32)).astype(f32))
    cr = ((fma(f32(0.5), r, np.full_like(r, f32(128.0))) + (f32(-0.4187) * g).astype(f32)).astype(f32) + (f32(-0.0813) * b).astype(f32)).astype(f32)
    for c in (cb, cr):
        assert c.min() >= 0.5 and c.max() <= 255.5      # positive: truncation == floor == clamp + truncate

