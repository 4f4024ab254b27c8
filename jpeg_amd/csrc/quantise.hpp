// quantise.hpp -- what the two encoders' quantisers share (kernels_encode.hip: the 8-bit fused encoder; kernels_generic.hip: any format):
// the rounding of the quotient and the packing of the quantised coefficients in zigzag pairs.  encode.swift:225-240.
#pragma once
#pragma clang fp contract(off)

#include "dct.hpp"

#include <cstdint>
#include <utility>

namespace jpeg_amd {

// horizontal frequency k of the coefficient at zigzag index z (inverse of zigzag_of over k), tabulated at compile time
struct ColumnOfZigzag {
    int k[64];
    constexpr ColumnOfZigzag() : k{}
    {
        for (int kk = 0; kk < 8; ++kk)
            for (int h = 0; h < 8; ++h) k[zigzag_of(kk, h)] = kk;
    }
};
constexpr ColumnOfZigzag kColumnOfZigzag{};
// the column after which the pair of zigzag slots (2m, 2m + 1) is complete
template <int M>
constexpr int pair_ready_after() { return kColumnOfZigzag.k[2 * M] > kColumnOfZigzag.k[2 * M + 1] ? kColumnOfZigzag.k[2 * M] : kColumnOfZigzag.k[2 * M + 1]; }

// copysign(pred(1/2), y) as ONE full-rate instruction: v_bitop3_b32 with the truth table "S0 ? S1 : S2" per bit (0xca) selects the magnitude's
// bits under the mask 0x7fffffff and y's sign bit elsewhere.  (The compiler's v_bfi_b32 for __builtin_copysignf issues at half the rate:
// tools/probe_rates3.hip, profiles/r06_probe_rates3.txt; bit-identical by construction, tools/probe_typed.hip T4.)
__device__ __forceinline__ float half_toward(float y)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_bitop3_b32(0x7fffffffu, __builtin_bit_cast(uint32_t, 0.49999997f), __builtin_bit_cast(uint32_t, y), 0xca));
}

// The pair of zigzag slots (2m, 2m + 1), truncated and packed by TWO conversions: v_cvt_i32_f32 writes the first integer, and the
// second conversion's SDWA destination select puts its low 16 bits into the upper half of the same register (dst_unused:
// UNUSED_PRESERVE) -- no v_cvt_pk_i16_i32 behind them (tools/probe_typed.hip T3: 2^20 random pairs).  |coefficient| < 2^15, so the
// low halves ARE the int16 values.  (Round 6: 5 % fewer issue cycles by the cost table and no measurable change of the kernel's
// time -- profiles/r06_ab_encode_quantiser.txt: the kernel is not bound by VALU issue alone.)
template <int M>
__device__ __forceinline__ void pack_pair_if_ready(const float (&zf)[64], uint32_t (&w)[32], int k)
{
    if (pair_ready_after<M>() == k) {   // k is a constant after unrolling
        asm volatile("v_cvt_i32_f32_e32 %0, %1\n\t"
                     "v_cvt_i32_f32_sdwa %0, %2 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD"
                     : "=&v"(w[M]) : "v"(zf[2 * M]), "v"(zf[2 * M + 1]));
    }
}
template <int... M>
__device__ __forceinline__ void pack_ready_pairs(const float (&zf)[64], uint32_t (&w)[32], int k, std::integer_sequence<int, M...>)
{
    (pack_pair_if_ready<M>(zf, w, k), ...);
}

// ... or written to LDS pair by pair (kernels_generic.hip: no 32 registers to spare): the pair (2m, 2m + 1) of a block whose 16-byte
// chunk c sits at slot c ^ swz
template <int M>
__device__ __forceinline__ void store_pair_if_ready(const float (&zf)[64], uint32_t *block, int swz, int k)
{
    if (pair_ready_after<M>() == k) {   // k is a constant after unrolling
        uint32_t w;
        asm volatile("v_cvt_i32_f32_e32 %0, %1\n\t"
                     "v_cvt_i32_f32_sdwa %0, %2 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD"
                     : "=&v"(w) : "v"(zf[2 * M]), "v"(zf[2 * M + 1]));
        block[4 * ((M >> 2) ^ swz) + (M & 3)] = w;
    }
}
template <int... M>
__device__ __forceinline__ void store_ready_pairs(const float (&zf)[64], uint32_t *block, int swz, int k, std::integer_sequence<int, M...>)
{
    (store_pair_if_ready<M>(zf, block, swz, k), ...);
}

}  // namespace jpeg_amd
