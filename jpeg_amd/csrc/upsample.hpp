// upsample.hpp -- the small exact building blocks the fused decode kernels share: centred 2x
// upsampling weights (decode.swift:4231-4251) and the byte <-> float conversions of the colour stage.
#pragma once
#pragma clang fp contract(off)

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace jpeg_amd {

// trunc(clamp(x, 0, 255)) == saturating round-to-nearest(x + kTruncBias) for the R and B
// channel values x = y + m c (see k_luma_fused and tests/test_colour_rounding.py)
constexpr float kTruncBias = -0.5f + 0.0009765625f;

template <int N>
__device__ __forceinline__ float ubyte(uint32_t v)
{
    return (float)((v >> (8 * N)) & 0xffu);  // v_cvt_f32_ubyteN
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
