/* jpeg_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.  See jpeg_oracle.h.
 *
 * Op-for-op scalar restatement of the reference's float32 arithmetic.  Every
 * statement below is one correctly rounded IEEE-754 binary32 operation in the
 * reference's order; compile with -ffp-contract=off (no FMA fusing).
 */
#include "jpeg_oracle.h"

#include <math.h>
#include <string.h>

/* decode.swift:1289-1298 */
int orc_zigzag(int x, int y)
{
    const int p = (x + y < 8) ? 1 : 0;
    const int q = (x + y) & 1;
    const int a = 72 * (p ^ 1);
    const int b = 2 * p - 1;
    const int n = b * (x + y) - 14 * p + 15;
    const int t = (n * (n + 1)) >> 1;
    return a + b * t - q * x - (q ^ 1) * y - 1;
}

/* decode.swift:3984-4017: row(h) = scale * Float(Q[k,h]);
 * result.h = h.h * v.h * row(h) with h.h[k] = r[k], v.h[k] = r[h];
 * Swift's `*` is left-associative: (r[k] * r[h]) * (scale * Float(Q)). */
void orc_modulate(const uint16_t q_zz[64], float scale, float out[64])
{
    static const float r[8] = {
        1.0f, 1.387039845f, 1.306562965f, 1.175875602f,
        1.0f, 0.785694958f, 0.541196100f, 0.275899379f};
    for (int h = 0; h < 8; ++h) {
        for (int k = 0; k < 8; ++k) {
            const float row = scale * (float)q_zz[orc_zigzag(k, h)];
            const float hv  = r[k] * r[h];
            out[8 * h + k]  = hv * row;
        }
    }
}

/* decode.swift:4042-4093.  One lane of the SIMD8 butterflies: h[0..7] are the
 * eight inputs along the transformed axis, g[0..7] the outputs. */
static inline void idct8(const float h[8], float shift, float g[8])
{
    const float a0 = shift + h[0] + h[4];
    const float a1 = shift + h[0] - h[4];
    const float b  = h[2] + h[6];
    const float c  = 1.414213562f * (h[2] - h[6]) - b;

    const float r0 = a0 + b;
    const float r1 = a1 + c;
    const float r2 = a1 - c;
    const float r3 = a0 - b;

    const float d0 = h[5] - h[3];
    const float d1 = h[1] + h[7];
    const float d2 = h[1] - h[7];
    const float d3 = h[5] + h[3];

    const float f  = 1.414213562f * (d1 - d3);
    const float l  = 1.847759065f * (d0 + d2);
    const float m0 = l - d2 * 1.082392200f;
    const float m1 = l - d0 * 2.613125930f;

    const float s0 = d1 + d3;
    const float s1 = m1 - s0;
    const float s2 = f  - s1;
    const float s3 = m0 - s2;

    g[0] = r0 + s0;
    g[1] = r1 + s1;
    g[2] = r2 + s2;
    g[3] = r3 + s3;
    g[4] = r3 - s3;
    g[5] = r2 - s2;
    g[6] = r1 - s1;
    g[7] = r0 - s0;
}

/* decode.swift:4020-4039 (load), :4095-4099 (idct8x8), :4107-4127 (store) */
static void idct_block(const int16_t *zz, const float q[64], float level,
                       float limit, uint16_t *out, size_t stride)
{
    /* zmap[8*h + k] = z(k, h); literal so concurrent callers need no init.
     * tests/test_oracle_unit.py checks it against orc_zigzag and against the
     * reference's own table (tests/unit/tests.swift:38-48). */
    static const unsigned char zmap[64] = {
         0,  1,  5,  6, 14, 15, 27, 28,
         2,  4,  7, 13, 16, 26, 29, 42,
         3,  8, 12, 17, 25, 30, 41, 43,
         9, 11, 18, 24, 31, 40, 44, 53,
        10, 19, 23, 32, 39, 45, 52, 54,
        20, 22, 33, 38, 46, 51, 55, 60,
        21, 34, 37, 47, 50, 56, 59, 61,
        35, 36, 48, 49, 57, 58, 62, 63};

    float in[8][8]; /* in[h][k] */
    for (int h = 0; h < 8; ++h)
        for (int k = 0; k < 8; ++k)
            in[h][k] = q[8 * h + k] * (float)zz[zmap[8 * h + k]];

    /* first pass: idct8 over tuple index h, per lane k; then transpose */
    float f[8][8]; /* f[k][y] after transpose */
    for (int k = 0; k < 8; ++k) {
        float col[8], res[8];
        for (int h = 0; h < 8; ++h) col[h] = in[h][k];
        idct8(col, 0.0f, res);
        for (int y = 0; y < 8; ++y) f[k][y] = res[y];
    }
    /* second pass: idct8 over tuple index k, per lane y; then transpose */
    for (int y = 0; y < 8; ++y) {
        float row[8], res[8];
        for (int k = 0; k < 8; ++k) row[k] = f[k][y];
        idct8(row, level, res);
        for (int x = 0; x < 8; ++x) {
            float v = res[x];
            /* SIMD.clamped(lowerBound:upperBound:) = min(max(v, lo), hi) */
            v = v > 0.0f ? v : 0.0f;
            v = v < limit ? v : limit;
            out[(size_t)y * stride + x] = (uint16_t)v; /* truncates */
        }
    }
}

void orc_idct_plane_rows(const int16_t *coef, int ux, int uy,
                         const uint16_t q_zz[64], int precision,
                         uint16_t *out, int by0, int by1)
{
    float q[64];
    orc_modulate(q_zz, 0.125f, q); /* scale: 0x1p-3, decode.swift:4107 */
    const size_t stride = (size_t)8 * ux;
    const float level   = ldexpf(1.0f, precision - 1) + 0.5f; /* :4110-4111 */
    const float limit   = ldexpf(1.0f, precision) - 1.0f;     /* :4112-4113 */
    if (by1 > uy) by1 = uy;
    for (int y = by0; y < by1; ++y)
        for (int x = 0; x < ux; ++x)
            idct_block(coef + 64 * ((size_t)ux * y + x), q, level, limit,
                       out + (size_t)8 * y * stride + 8 * x, stride);
}

void orc_idct_plane(const int16_t *coef, int ux, int uy,
                    const uint16_t q_zz[64], int precision, uint16_t *out)
{
    /* force zigzag LUT init before any threaded use */
    orc_idct_plane_rows(coef, ux, uy, q_zz, precision, out, 0, uy);
}

/* decode.swift:4182-4276 */
void orc_interleave_rows(const uint16_t *const *planes, const int *ux,
                         const int *uy, const int *fx, const int *fy,
                         int count, int sx, int sy, int W, int H, int cosited,
                         uint16_t *out, int y0, int y1)
{
    if (y1 > H) y1 = H;
    if (count == 1) {
        /* :4185-4197 crop copy */
        const size_t pw = (size_t)8 * ux[0];
        for (int y = y0; y < y1; ++y)
            for (int x = 0; x < W; ++x)
                out[(size_t)y * W + x] = planes[0][x + pw * y];
        return;
    }
    for (int p = 0; p < count; ++p) {
        const uint16_t *plane = planes[p];
        const size_t pw = (size_t)8 * ux[p];
        const int ph = 8 * uy[p];
        if (fx[p] == sx && fy[p] == sy) {
            /* :4206-4215 fast path */
            for (int y = y0; y < y1; ++y)
                for (int x = 0; x < W; ++x)
                    out[((size_t)y * W + x) * count + p] = plane[x + pw * y];
            continue;
        }
        int ax, ay, bx, by, cx, cy;
        if (cosited) {
            ax = 0; ay = 0; bx = fx[p]; by = fy[p]; cx = sx; cy = sy;
        } else {
            ax = fx[p] - sx;  ay = fy[p] - sy;
            bx = 2 * fx[p];   by = 2 * fy[p];
            cx = 2 * sx;      cy = 2 * sy;
        }
        const int dx = (int)pw - 1, dy = ph - 1;
        for (int y = y0; y < y1; ++y) {
            /* C integer / and % truncate toward zero like quotientAndRemainder */
            const int iy = (ay + by * y) / cy, gy = (ay + by * y) % cy;
            const int jy = iy + 1 < dy ? iy + 1 : dy;
            float ty = (float)gy / (float)cy;
            ty = ty < 1.0f ? ty : 1.0f;
            ty = ty > 0.0f ? ty : 0.0f;
            for (int x = 0; x < W; ++x) {
                const int ix = (ax + bx * x) / cx, gx = (ax + bx * x) % cx;
                const int jx = ix + 1 < dx ? ix + 1 : dx;
                float tx = (float)gx / (float)cx;
                tx = tx < 1.0f ? tx : 1.0f;
                tx = tx > 0.0f ? tx : 0.0f;
                const float u00 = (float)plane[ix + pw * iy];
                const float u01 = (float)plane[jx + pw * iy];
                const float u10 = (float)plane[ix + pw * jy];
                const float u11 = (float)plane[jx + pw * jy];
                const float v0 = u00 * (1.0f - tx) + u01 * tx;
                const float v1 = u10 * (1.0f - tx) + u11 * tx;
                const float w  = v0 * (1.0f - ty) + v1 * ty;
                out[((size_t)y * W + x) * count + p] = (uint16_t)roundf(w);
            }
        }
    }
}

void orc_interleave(const uint16_t *const *planes, const int *ux,
                    const int *uy, const int *fx, const int *fy, int count,
                    int sx, int sy, int W, int H, int cosited, uint16_t *out)
{
    orc_interleave_rows(planes, ux, uy, fx, fy, count, sx, sy, W, H, cosited,
                        out, 0, H);
}

/* jpeg.swift:343-354: T(max(T.min, min(x, T.max))) -- truncating conversion */
static inline uint8_t clamp_u8(float x)
{
    x = x < 255.0f ? x : 255.0f;
    x = x > 0.0f ? x : 0.0f;
    return (uint8_t)x;
}

/* jpeg.swift:441-453: x = (Float(y) + m_cb * (Float(cb) - 128)) + m_cr * (Float(cr) - 128) */
static inline void ycc_to_rgb(uint8_t y, uint8_t cb, uint8_t cr, uint8_t *out)
{
    const float fy = (float)y;
    const float b  = (float)cb - 128.0f;
    const float r  = (float)cr - 128.0f;
    out[0] = clamp_u8((fy + 0.00000f * b) + 1.40200f * r);
    out[1] = clamp_u8((fy + -0.34414f * b) + -0.71414f * r);
    out[2] = clamp_u8((fy + 1.77200f * b) + 0.00000f * r);
}

/* jpeg.swift:551-572 (UInt8(UInt16) conversions there trap on overflow; the
 * reference never reaches that because idct clamps to 2^P - 1 with P = 8). */
void orc_unpack_rgb8(const uint16_t *in, size_t npx, int ncomp, uint8_t *out)
{
    if (ncomp == 1) {
        for (size_t i = 0; i < npx; ++i)
            ycc_to_rgb((uint8_t)in[i], 128, 128, out + 3 * i);
    } else {
        for (size_t i = 0; i < npx; ++i)
            ycc_to_rgb((uint8_t)in[3 * i], (uint8_t)in[3 * i + 1],
                       (uint8_t)in[3 * i + 2], out + 3 * i);
    }
}

/* jpeg.swift:493-514 */
void orc_unpack_ycc8(const uint16_t *in, size_t npx, int ncomp, uint8_t *out)
{
    if (ncomp == 1) {
        for (size_t i = 0; i < npx; ++i) {
            out[3 * i]     = (uint8_t)in[i];
            out[3 * i + 1] = 128;
            out[3 * i + 2] = 128;
        }
    } else {
        for (size_t i = 0; i < 3 * npx; ++i) out[i] = (uint8_t)in[i];
    }
}

/* jpeg.swift:463-478: x = ((m0 + m_r * r) + m_g * g) + m_b * b */
static inline void rgb_to_ycc(const uint8_t *rgb, uint8_t *out)
{
    const float r = (float)rgb[0], g = (float)rgb[1], b = (float)rgb[2];
    out[0] = clamp_u8(((0.0f   +  0.2990f * r) +  0.5870f * g) +  0.1140f * b);
    out[1] = clamp_u8(((128.0f + -0.1687f * r) + -0.3313f * g) +  0.5000f * b);
    out[2] = clamp_u8(((128.0f +  0.5000f * r) + -0.4187f * g) + -0.0813f * b);
}

/* jpeg.swift:584-599 */
void orc_pack_rgb8(const uint8_t *in, size_t npx, int ncomp, uint16_t *out)
{
    for (size_t i = 0; i < npx; ++i) {
        uint8_t ycc[3];
        rgb_to_ycc(in + 3 * i, ycc);
        if (ncomp == 1) {
            out[i] = ycc[0];
        } else {
            out[3 * i]     = ycc[0];
            out[3 * i + 1] = ycc[1];
            out[3 * i + 2] = ycc[2];
        }
    }
}

/* jpeg.swift:527-539 */
void orc_pack_ycc8(const uint8_t *in, size_t npx, int ncomp, uint16_t *out)
{
    if (ncomp == 1) {
        for (size_t i = 0; i < npx; ++i) out[i] = in[3 * i];
    } else {
        for (size_t i = 0; i < 3 * npx; ++i) out[i] = in[i];
    }
}

/* encode.swift:389-425 */
void orc_decompose_plane(const uint16_t *in, int W, int H, int count, int p,
                         int fx, int fy, int sx, int sy, int ux, int uy,
                         uint16_t *out)
{
    const int rx = sx / fx, ry = sy / fy;            /* response, :403 */
    const float magnitude = (float)(rx * ry);        /* :404 */
    const int pw = 8 * ux, ph = 8 * uy;
    for (int y = 0; y < ph; ++y) {
        for (int x = 0; x < pw; ++x) {
            const int bx = x * sx / fx, by = y * sy / fy; /* :407-411 */
            long sum = 0;
            for (int yy = by; yy < by + ry; ++yy) {
                const int iy = yy < H - 1 ? yy : H - 1;
                for (int xx = bx; xx < bx + rx; ++xx) {
                    const int ix = xx < W - 1 ? xx : W - 1;
                    sum += in[((size_t)W * iy + ix) * count + p];
                }
            }
            out[(size_t)pw * y + x] = (uint16_t)((float)sum / magnitude);
        }
    }
}

/* encode.swift:123-188 */
static inline void fdct8(const float g[8], float shift, float out[8])
{
    const float a0 = g[0] + g[7];
    const float a1 = g[1] + g[6];
    const float a2 = g[2] + g[5];
    const float a3 = g[3] + g[4];

    const float b0 = a0 + a3;
    const float b1 = a1 + a2;
    const float b2 = a1 - a2;
    const float b3 = a0 - a3;

    const float c  = 0.707106781f * (b2 + b3);
    const float r0 = b0 + b1 - shift;
    const float r1 = b3 + c;
    const float r2 = b0 - b1;
    const float r3 = b3 - c;

    const float d0 = g[3] - g[4];
    const float d1 = g[2] - g[5];
    const float d2 = g[1] - g[6];
    const float d3 = g[0] - g[7];

    const float f0 = d0 + d1;
    const float f1 = d1 + d2;
    const float f2 = d2 + d3;

    const float k  = 0.707106781f * f1;
    const float l  = 0.382683433f * (f0 - f2);
    const float m0 = l + f0 * 0.541196100f;
    const float m1 = l + f2 * 1.306562965f;

    const float n0 = d3 + k;
    const float n1 = d3 - k;

    const float s0 = n0 + m1;
    const float s1 = n1 - m0;
    const float s2 = n1 + m0;
    const float s3 = n0 - m1;

    out[0] = r0; out[1] = s0; out[2] = r1; out[3] = s1;
    out[4] = r2; out[5] = s2; out[6] = r3; out[7] = s3;
}

/* encode.swift:80-99 (load), :191-196 (fdct8x8), :219-241 (quantise+scatter) */
static void fdct_block(const uint16_t *src, size_t stride, const float q[64],
                       float level, float limit, int16_t *zz)
{
    float g[8][8]; /* g[y][x] */
    for (int y = 0; y < 8; ++y)
        for (int x = 0; x < 8; ++x) {
            const float v = (float)src[(size_t)y * stride + x];
            g[y][x] = limit < v ? limit : v; /* pointwiseMin(limit, v) */
        }
    /* f = fdct8(transpose(g), level): tuple index = x, lane = y */
    float f[8][8]; /* f[k][y] */
    for (int y = 0; y < 8; ++y) {
        float res[8];
        fdct8(g[y], level, res);
        for (int k = 0; k < 8; ++k) f[k][y] = res[k];
    }
    /* h = fdct8(transpose(f), 0): tuple index = y, lane = k */
    for (int k = 0; k < 8; ++k) {
        float res[8];
        fdct8(f[k], 0.0f, res);
        for (int h = 0; h < 8; ++h) {
            const float v = res[h] / q[8 * h + k];
            /* SIMD8<Int16>(v, rounding: .toNearestOrAwayFromZero) */
            zz[orc_zigzag(k, h)] = (int16_t)roundf(v);
        }
    }
}

void orc_fdct_plane_rows(const uint16_t *plane, int ux, int uy,
                         const uint16_t q_zz[64], int precision,
                         int16_t *coef, int by0, int by1)
{
    float q[64];
    orc_modulate(q_zz, 8.0f, q);                           /* :205-209 */
    const float level = ldexpf(1.0f, precision - 1) * 8.0f; /* :215-216 */
    const float limit = ldexpf(1.0f, precision) - 1.0f;     /* :217-218 */
    const size_t stride = (size_t)8 * ux;
    if (by1 > uy) by1 = uy;
    for (int y = by0; y < by1; ++y)
        for (int x = 0; x < ux; ++x)
            fdct_block(plane + (size_t)8 * y * stride + 8 * x, stride, q,
                       level, limit, coef + 64 * ((size_t)ux * y + x));
}

void orc_fdct_plane(const uint16_t *plane, int ux, int uy,
                    const uint16_t q_zz[64], int precision, int16_t *coef)
{
    orc_fdct_plane_rows(plane, ux, uy, q_zz, precision, coef, 0, uy);
}

/* encode.swift:260-333 */
void orc_compression_quanta(int kind, double t, uint16_t out_zz[64])
{
    static const uint16_t lum[64] = {
        16, 11, 10, 16, 124, 140, 151, 161,
        12, 12, 14, 19, 126, 158, 160, 155,
        14, 13, 16, 24, 140, 157, 169, 156,
        14, 17, 22, 29, 151, 187, 180, 162,
        18, 22, 37, 56, 168, 109, 103, 177,
        24, 35, 55, 64, 181, 104, 113, 192,
        49, 64, 78, 87, 103, 121, 120, 101,
        72, 92, 95, 98, 112, 100, 103, 199};
    static const uint16_t chr[64] = {
        17, 18, 24, 47, 99, 99, 99, 99,
        18, 21, 26, 66, 99, 99, 99, 99,
        24, 26, 56, 99, 99, 99, 99, 99,
        47, 66, 99, 99, 99, 99, 99, 99,
        99, 99, 99, 99, 99, 99, 99, 99,
        99, 99, 99, 99, 99, 99, 99, 99,
        99, 99, 99, 99, 99, 99, 99, 99,
        99, 99, 99, 99, 99, 99, 99, 99};
    const uint16_t *key = kind == 0 ? lum : chr;
    for (int h = 0; h < 8; ++h)
        for (int k = 0; k < 8; ++k) {
            double v = round(1.0 * (1 - t) + (double)key[8 * h + k] * t);
            v = v < 255.0 ? v : 255.0;
            v = v > 1.0 ? v : 1.0;
            out_zz[orc_zigzag(k, h)] = (uint16_t)v;
        }
}
