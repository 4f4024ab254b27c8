// dct.hpp -- device-side 8-point butterflies, zigzag map and table modulation.
//
// Arithmetic contract: every statement is ONE IEEE-754 binary32
// operation in the reference's order.  Translation units including this header MUST be
// compiled with -ffp-contract=off; the pragma below is a second line of defence.
//
// Reference: tayloraswift/jpeg @ 2024_08_07, sources/jpeg/decode.swift, encode.swift.
#pragma once
#pragma clang fp contract(off)

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace jpeg_amd {

// natural index (8*h + k) -> zigzag index; tabulation of Table.Quantization.z(k:h:)
// (decode.swift:1289-1298; known-answer table tests/unit/tests.swift:38-48).
__host__ __device__ constexpr int zigzag_of(int k, int h)
{
    const int p = (k + h < 8) ? 1 : 0;
    const int q = (k + h) & 1;
    const int a = 72 * (p ^ 1);
    const int b = 2 * p - 1;
    const int n = b * (k + h) - 14 * p + 15;
    const int t = (n * (n + 1)) >> 1;
    return a + b * t - q * k - (q ^ 1) * h - 1;
}

// Spectral.Plane.modulate(quanta:scale:) -- decode.swift:3984-4017.
// q[h][k] = (r[k] * r[h]) * (scale * Float(Q[z(k,h)])), left-associative like Swift's `*`.
__host__ __device__ inline float modulate_entry(int k, int h, float scale, uint16_t quantum)
{
    const float r[8] = {1.0f, 1.387039845f, 1.306562965f, 1.175875602f,
                        1.0f, 0.785694958f, 0.541196100f, 0.275899379f};
    const float hv  = r[k] * r[h];
    const float row = scale * (float)quantum;
    return hv * row;
}

// idct8 -- decode.swift:4042-4093.  h[0..7] along the transformed axis.
// SHIFTED = false is the first pass (shift: 0): the reference's `0 + h0` only ever
// changes the sign of a zero, which no later operation of the path can observe.
template <bool SHIFTED>
__device__ __forceinline__ void idct8(const float (&h)[8], float shift, float (&g)[8])
{
    const float e  = SHIFTED ? shift + h[0] : h[0];
    const float a0 = e + h[4];
    const float a1 = e - h[4];
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
    const float s2 = f - s1;
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

// fdct8 -- encode.swift:123-188.  Output order (r0, s0, r1, s1, r2, s2, r3, s3).
// SHIFTED = false is the second pass (shift: 0): `x - 0` is exact.
template <bool SHIFTED>
__device__ __forceinline__ void fdct8(const float (&g)[8], float shift, float (&o)[8])
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
    const float r0 = SHIFTED ? (b0 + b1) - shift : b0 + b1;
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

    o[0] = r0; o[1] = s0; o[2] = r1; o[3] = s1;
    o[4] = r2; o[5] = s2; o[6] = r3; o[7] = s3;
}

// One quantised coefficient out of a block held as 32 packed dwords (zigzag order).
__device__ __forceinline__ float coef_as_float(const uint32_t (&w)[32], int z)
{
    const int32_t word = (int32_t)w[z >> 1];
    const int32_t c    = (z & 1) ? (word >> 16) : (int32_t)(int16_t)word;
    return (float)c;
}

// A modulated table stored TRANSPOSED in LDS (t[8 k + h]): the first pass reads the eight entries of one column k
// (h = 0..7) together, and transposed they are 32 contiguous bytes = two ds_read_b128 instead of four ds_read2_b32.
// Indexed like the natural table (q[8 h + k]), so idct_block & co. take it unchanged.
struct TransposedTable {
    const float *t;
    __device__ __forceinline__ float operator[](int i) const { return t[8 * (i & 7) + (i >> 3)]; }
    __device__ __forceinline__ TransposedTable operator+(int k) const = delete;
};

// Spectral.Plane.load + idct8x8 -- decode.swift:4020-4039, 4095-4099.
// w: 64 int16 in zigzag order; q: modulated table, natural order q[8*h + k] (any address
// space, uniform across the wave); g[8*y + x]: samples before clamp (level already added).
template <typename QPtr>
__device__ __forceinline__ void idct_block(const uint32_t (&w)[32], QPtr q, float level,
                                           float (&g)[64])
{
    float f[64];  // f[8*k + y]: after first pass + transpose
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        float h[8], res[8];
#pragma unroll
        for (int hh = 0; hh < 8; ++hh)
            h[hh] = q[8 * hh + k] * coef_as_float(w, zigzag_of(k, hh));
        idct8<false>(h, 0.0f, res);
#pragma unroll
        for (int y = 0; y < 8; ++y) f[8 * k + y] = res[y];
    }
#pragma unroll
    for (int y = 0; y < 8; ++y) {
        float r[8], res[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) r[k] = f[8 * k + y];
        idct8<true>(r, level, res);
#pragma unroll
        for (int x = 0; x < 8; ++x) g[8 * y + x] = res[x];
    }
}

// The same two passes, split so that a kernel can interleave other work between them and
// produce output rows one at a time (idct_block == idct_pass1 + 8 x idct_pass2_row).
template <typename QPtr>
__device__ __forceinline__ void idct_pass1(const uint32_t (&w)[32], QPtr q, float (&f)[64])
{
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        float h[8], res[8];
#pragma unroll
        for (int hh = 0; hh < 8; ++hh)
            h[hh] = q[8 * hh + k] * coef_as_float(w, zigzag_of(k, hh));
        idct8<false>(h, 0.0f, res);
#pragma unroll
        for (int y = 0; y < 8; ++y) f[8 * k + y] = res[y];
    }
}

__device__ __forceinline__ void idct_pass2_row(const float (&f)[64], int y, float level,
                                               float (&g)[8])
{
    float r[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) r[k] = f[8 * k + y];
    idct8<true>(r, level, g);
}

// x with its sign flipped where `flip` is 0x80000000 (else 0): r0 + flip_sign(s0, ...) is idct8's output 0 (r0 + s0) or 7
// (r0 - s0 == r0 + (-s0), the same IEEE operation) chosen PER LANE without a select -- v_cndmask_b32 issues ten times slower
// than a v_xor_b32 on gfx950 (profiles/r05_probe_valu_classes.txt).
__device__ __forceinline__ float flip_sign(float x, uint32_t flip)
{
    return __builtin_bit_cast(float, __builtin_bit_cast(uint32_t, x) ^ flip);
}

// ONE output row of idct_block: row 0 (flip == 0) or row 7 (flip == 0x80000000; may differ per lane).
// Same operations in the same order for the values that are kept; what the other rows would have
// needed is never computed: of the first pass only idct8's r0 = (h0 + h4) + (h2 + h6) and s0 = (h1 + h7) + (h5 + h3).
template <typename QPtr>
__device__ __forceinline__ void idct_block_edge_row(const uint32_t (&w)[32], QPtr q, float level, uint32_t flip,
                                                    float (&g)[8])
{
    float t[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        float h[8];
#pragma unroll
        for (int hh = 0; hh < 8; ++hh)
            h[hh] = q[8 * hh + k] * coef_as_float(w, zigzag_of(k, hh));
        const float a0 = h[0] + h[4];          // idct8<false>: e = h0
        const float b  = h[2] + h[6];
        const float r0 = a0 + b;
        const float d1 = h[1] + h[7];
        const float d3 = h[5] + h[3];
        const float s0 = d1 + d3;
        t[k] = r0 + flip_sign(s0, flip);       // res[0] = r0 + s0, res[7] = r0 - s0
    }
    idct8<true>(t, level, g);
}

// The first and the last COLUMN of idct_block (samples x = 0 and x = 7 of all eight rows): the first pass in full, of the
// second pass only what outputs 0 and 7 need -- the same operations in the same order (idct8: r0 + s0, r0 - s0).
template <typename QPtr>
__device__ __forceinline__ void idct_block_edge_cols(const uint32_t (&w)[32], QPtr q, float level,
                                                     float (&c0)[8], float (&c7)[8])
{
    float f[64];  // f[8*k + y]
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        float h[8], res[8];
#pragma unroll
        for (int hh = 0; hh < 8; ++hh)
            h[hh] = q[8 * hh + k] * coef_as_float(w, zigzag_of(k, hh));
        idct8<false>(h, 0.0f, res);
#pragma unroll
        for (int y = 0; y < 8; ++y) f[8 * k + y] = res[y];
    }
#pragma unroll
    for (int y = 0; y < 8; ++y) {
        const float e  = level + f[8 * 0 + y];
        const float a0 = e + f[8 * 4 + y];
        const float b  = f[8 * 2 + y] + f[8 * 6 + y];
        const float r0 = a0 + b;
        const float d1 = f[8 * 1 + y] + f[8 * 7 + y];
        const float d3 = f[8 * 5 + y] + f[8 * 3 + y];
        const float s0 = d1 + d3;
        c0[y] = r0 + s0;
        c7[y] = r0 - s0;
    }
}

// ONE edge column of idct_block, chosen per lane: column 0 (flip == 0) or column 7 (flip == 0x80000000) -- idct_block_edge_cols
// with the final r0 +- s0 folded into one addition (flip_sign above).
template <typename QPtr>
__device__ __forceinline__ void idct_block_edge_col(const uint32_t (&w)[32], QPtr q, float level, uint32_t flip, float (&c)[8])
{
    float f[64];  // f[8*k + y]
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        float h[8], res[8];
#pragma unroll
        for (int hh = 0; hh < 8; ++hh)
            h[hh] = q[8 * hh + k] * coef_as_float(w, zigzag_of(k, hh));
        idct8<false>(h, 0.0f, res);
#pragma unroll
        for (int y = 0; y < 8; ++y) f[8 * k + y] = res[y];
    }
#pragma unroll
    for (int y = 0; y < 8; ++y) {
        const float e  = level + f[8 * 0 + y];
        const float a0 = e + f[8 * 4 + y];
        const float b  = f[8 * 2 + y] + f[8 * 6 + y];
        const float r0 = a0 + b;
        const float d1 = f[8 * 1 + y] + f[8 * 7 + y];
        const float d3 = f[8 * 5 + y] + f[8 * 3 + y];
        const float s0 = d1 + d3;
        c[y] = r0 + flip_sign(s0, flip);
    }
}

// ---- the edge column of a block by (block, column) work-items ------------------------------------------------------------------
// idct_block_edge_cols spends a whole wave pass on every block it is given; where a pass holds only a handful of blocks (the 8
// neighbour blocks left and right of a 4:2:2 strip) 56 of 64 work-items idle through ~800 instructions.  Here EIGHT consecutive
// lanes share one block: lane j of the group runs the first pass of ONE column -- k = edge_col_of_lane(j): dequantise + idct8
// over the vertical frequency, the same 8 + 34 operations idct_block spends on that column -- and the second pass's outputs 0
// and 7 of every row are formed ACROSS the eight lanes with three levels of DPP adds, in the reference's own association
// (idct8<true>: e = level + f0; a0 = e + f4; b = f2 + f6; r0 = a0 + b; d1 = f1 + f7; d3 = f5 + f3; s0 = d1 + d3; g0 = r0 + s0;
// g7 = r0 - s0 -- float addition is commutative, so which lane of a pair holds which operand does not matter):
//     lanes j = 0..7 hold columns k = 0, 2, 1, 5, 4, 6, 7, 3
//     level 1  lane j += lane j + 4   (row_shl:4)            lanes 0..3: a0, b, d1, d3
//     level 2  lane j += lane j ^ 1   (quad_perm [1,0,3,2])  lane 0: r0, lane 2: s0
//     level 3  lane 0 +- lane 2       (quad_perm [2,3,0,1])  lane 0: g0 or g7
// ~130 instructions per wave for 8 blocks instead of ~810.  Lane 0 of each group ends up with the column; the other lanes'
// results are by-products and must be ignored.
__device__ __forceinline__ int edge_col_of_lane(int j) { return (int)((0x37645120u >> (4 * j)) & 7u); }

__device__ __forceinline__ float dpp_f32(float v, int ctrl_row_shl4_quad1032_quad2301)
{
    const int i = __builtin_bit_cast(int, v);
    int r;
    if (ctrl_row_shl4_quad1032_quad2301 == 0) r = __builtin_amdgcn_update_dpp(i, i, 0x104, 0xf, 0xf, false);        // row_shl:4
    else if (ctrl_row_shl4_quad1032_quad2301 == 1) r = __builtin_amdgcn_update_dpp(i, i, 0xB1, 0xf, 0xf, false);   // quad_perm [1,0,3,2]
    else r = __builtin_amdgcn_update_dpp(i, i, 0x4E, 0xf, 0xf, false);                                             // quad_perm [2,3,0,1]
    return __builtin_bit_cast(float, r);
}

// coef[h], q[h]: the quantised coefficients (k, h) and the modulated table entries q[8 h + k] of THIS lane's column k =
// edge_col_of_lane(j), j = lane & 7; first: the block's column 0 is wanted (else column 7) -- uniform over the group.
// edge[y]: sample (x = 0 or 7, row y) before the clamp, valid in the group's lane 0.
__device__ __forceinline__ void idct_edge_col_split(const int (&coef)[8], const float (&q)[8], float level, bool first, int j,
                                                    float (&edge)[8])
{
#ifdef JA_DEBUG_ASSERTS   // the precondition, checked in debug builds: all 64 lanes active (a DPP read of an inactive lane keeps the old value)
    if (__builtin_amdgcn_read_exec() != ~0ull) __builtin_trap();
#endif
    float h[8], f[8];
#pragma unroll
    for (int hh = 0; hh < 8; ++hh) h[hh] = q[hh] * (float)coef[hh];
    idct8<false>(h, 0.0f, f);                      // this lane's column after the first pass: f[y]
    const float lv = j == 0 ? level : -0.0f;       // (-0) + f == f for every f, signed zeros included
#pragma unroll
    for (int y = 0; y < 8; ++y) {
        float x = lv + f[y];                       // lane 0: e = level + f0
        x = x + dpp_f32(x, 0);                     // a0 = e + f4 | b = f2 + f6 | d1 = f1 + f7 | d3 = f5 + f3
        x = x + dpp_f32(x, 1);                     // r0 = a0 + b | . | s0 = d1 + d3 | .
        const float s0 = dpp_f32(x, 2);
        edge[y] = first ? x + s0 : x - s0;         // g0 = r0 + s0, g7 = r0 - s0
    }
}

// Planar.Plane.load + fdct8x8 -- encode.swift:80-99, 191-196.
// g[8*y + x]: samples already min(limit, Float(sample)); out H[8*h + k] before quantise.
__device__ __forceinline__ void fdct_block(const float (&g)[64], float level, float (&H)[64])
{
    float f[64];  // f[8*k + y]
#pragma unroll
    for (int y = 0; y < 8; ++y) {
        float r[8], res[8];
#pragma unroll
        for (int x = 0; x < 8; ++x) r[x] = g[8 * y + x];
        fdct8<true>(r, level, res);
#pragma unroll
        for (int k = 0; k < 8; ++k) f[8 * k + y] = res[k];
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        float r[8], res[8];
#pragma unroll
        for (int y = 0; y < 8; ++y) r[y] = f[8 * k + y];
        fdct8<false>(r, 0.0f, res);
#pragma unroll
        for (int h = 0; h < 8; ++h) H[8 * h + k] = res[h];
    }
}

// clamp to [0, limit] then truncate toward zero (decode.swift:4121-4122;
// SIMD.clamped + SIMD8<UInt16>(_:) ).  Inputs are finite, so med3 == min(max()).
__device__ __forceinline__ uint32_t clamp_trunc(float v, float limit)
{
    return (uint32_t)__builtin_amdgcn_fmed3f(v, 0.0f, limit);
}

// Float.rounded() / rounding: .toNearestOrAwayFromZero -- exactly, for either sign, and without a select: with r = trunc(v) the
// difference d = v - r is exact and lies in (-1, 1), 2 d is exact, and trunc(2 d) is +-1 exactly when |d| >= 1/2 (0 otherwise);
// |v| >= 2^23 is its own truncation.  (roundf() compiles to a sequence around v_cndmask_b32, which issues ten times slower
// than anything else on gfx950: profiles/r05_probe_valu_classes.txt.)
__device__ __forceinline__ float round_half_away(float v)
{
    const float r = __builtin_truncf(v);
    const float d = v - r;
    return r + __builtin_truncf(d + d);
}

}  // namespace jpeg_amd
