// upsample.hpp -- the small exact building blocks the fused decode kernels share: centred 2x
// upsampling weights (decode.swift:4231-4251) and the byte <-> float conversions of the colour stage.
#pragma once
#pragma clang fp contract(off)

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace jpeg_amd {

template <int N>
__device__ __forceinline__ float ubyte(uint32_t v)
{
    return (float)((v >> (8 * N)) & 0xffu);  // v_cvt_f32_ubyteN
}

// Rounding without a floor (the 4:2:0 stack walk, kernels_quad.hip).  The final chroma value of a pixel is
// floor(v / 16 + 1/2) [- 128] with v = 9a + 3b + 3c + d an integer below 2^12.  A byte p enters the arithmetic as the float
// P = 2^15 + p + 1/32, built by ONE v_perm_b32 from the constant 0x47000008 (exponent of 2^15; mantissa = p << 8 | 8) -- the
// same cost as v_cvt_f32_ubyte.  Then 3 P + P' = 2^17 + (3p + p') + 1/8 and 3 H + H' = 2^19 + v + 1/2, both exact (21
// significant bits), and
//     t = fma(2^19 + v + 1/2, 1/16, C),  C = 1.5 * 2^23 - 2^15 [- 128]
// is ONE rounding of  1.5 * 2^23 [- 128] + v / 16 + 1/32  to a float whose ulp is 1: v / 16 + 1/32 is never half-way
// (its fraction is k/16 + 1/32), so t = 1.5 * 2^23 + floor(v / 16 + 1/2) [- 128], and t - 1.5 * 2^23 is the value itself.
// An FMA and a subtraction (full-rate) instead of an FMA and a floor (quarter-rate class): -128 slow instructions per
// block.  tests/test_colour_rounding.py enumerates every v and every byte pair.
constexpr uint32_t kBytePerm = 0x47000008u;             // bytes 3, 2, 0 of P (byte 1 is the sample)
constexpr float kMagic = 12582912.0f;                   // 1.5 * 2^23
template <int N>
__device__ __forceinline__ float ubyte_magic(uint32_t v)
{
    // D.byte3 = C.byte3, D.byte2 = C.byte2, D.byte1 = v.byte N, D.byte0 = C.byte0   (S0 = v: selectors 4..7, S1 = C: 0..3)
    return __uint_as_float(__builtin_amdgcn_perm(v, kBytePerm, 0x03020000u | ((4u + N) << 8)));
}

// 3a + b, exact (small integers): one v_fma_f32
__device__ __forceinline__ float w31(float a, float b) { return __builtin_fmaf(a, 3.0f, b); }

// one row of 2x-upsampled weights from 6 neighbours p[0..5] (p[0] = sample left of the
// block's first chroma sample): out[x] = 4 * bilinear value
__device__ __forceinline__ void lerp_row_2x(const float (&p)[6], float (&o)[8])
{
    o[0] = w31(p[1], p[0]); o[1] = w31(p[1], p[2]);
    o[2] = w31(p[2], p[1]); o[3] = w31(p[2], p[3]);
    o[4] = w31(p[3], p[2]); o[5] = w31(p[3], p[4]);
    o[6] = w31(p[4], p[3]); o[7] = w31(p[4], p[5]);
}

}  // namespace jpeg_amd
