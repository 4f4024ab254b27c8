// kernels_encode.hip -- fused pixels -> Spectral fast path for the built-in 8-bit formats.
//
// Replaces Rectangular.pack(...) -> decomposed() -> Planar.fdct(quanta:) (encode.swift:453,
// 389, 353) for ycc8 images whose luma has the full sampling factor and whose chroma planes
// are subsampled 1x or 2x per axis, and for y8 images, without materialising Rectangular /
// Planar in HBM: one kernel, RGB8 (or YCbCr8) bytes in, quantised zigzag coefficients out.
//
// One workgroup = one tile of 32 x 16 luma blocks (256 x 128 px); a work-item converts and
// transforms two luma blocks (rows t/32 and t/32 + 8 of the tile), pools their chroma into an
// LDS tile (box filter of encode.swift:402-423: the window of a 2x subsampled sample lies
// inside one 8x8 luma block, so no halo is needed), and after one barrier transforms its
// share of the tile's chroma blocks (one per work-item for 4:2:0).
//
// Arithmetic is the reference's, op for op (-ffp-contract=off): colour matrix
// jpeg.swift:463-478, FDCT encode.swift:123-196, true IEEE division by the modulated table and
// round-half-away (encode.swift:225-240).  Exact simplifications:
//   - `pointwiseMin(limit, v)` is the identity for 8-bit samples (v <= 255 = limit);
//   - the colour results need no clamp before truncation: Y in [0, 255.0001], Cb/Cr in
//     [0.5, 255.5], so floor() alone is clamp + truncate;
//   - Float(sum) / Float(n) truncated, n in {1, 2, 4}, is floor(sum * (1/n)) exactly.
//
// Development switches (tools/build_exp.sh, never in the product build): JA_X_ENC_NOSTORE,
// JA_X_ENC_L2LOAD -- the kernel without its stores / with every load hitting L2; JA_X_ENC_NOCOMPUTE -- its memory traffic with next to no
// arithmetic (round 6, profiles/r06_ablate_encode.txt: memory alone and arithmetic alone need the same 71 us at 8192 x 8192, the kernel 93);
// JA_X_ENC_NOCHROMA -- without the two-wave
// chroma tail of a tile (round 4: 23.6 -> 19.3 us at 4096 x 4096, 86.1 -> 68.7 at 8192 x 8192: the tail's share of the
// tile's instructions, 18 %, is its share of the time -- the kernel is bound by the instructions it issues in all, not by
// the longest wave); JA_X_ENC_TY, JA_X_ENC_PERHALF_WAVES.
#pragma clang fp contract(off)

#include "dct.hpp"
#include "quantise.hpp"
#include "fused_common.hpp"   // trunc_pack16 (clamp + truncate + pack under round-toward-zero), kThreads
#include "kernels.hpp"

#include <cstdlib>
#include <utility>

typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));

namespace jpeg_amd {

namespace {

constexpr int ETX = 32;  // luma blocks per tile row
constexpr int ETY = 16;  // luma blocks per tile column (8 for the small-image variant, template parameter TY)

struct EncArgs {
    const uint8_t *px;
    size_t px_stride;
    int W, H;
    int16_t *coef[3];
    size_t coef_stride[3];
    int ux[3], uy[3];
    const uint16_t *quanta;
    size_t quanta_stride;
    int qi[3];
    int tiles_x;
};

template <int N>
__device__ __forceinline__ float ubyte(uint32_t v)
{
    return (float)((v >> (8 * N)) & 0xffu);  // v_cvt_f32_ubyteN
}

// RGB.ycc -- jpeg.swift:463-478: x = ((m0 + m_r r) + m_g g) + m_b b, then clamp + truncate.
// The two products by 0.5 are exact (r, b are bytes), so `x + 0.5 b` and `128 + 0.5 r` round once either way: those two
// steps are one FMA each.  No other step may be fused: every other FMA placement changes some of the 2^24 results
// (tests/test_colour_rounding.py enumerates them).
// RAWC: Cb / Cr are returned BEFORE their truncation (they lie in [0.5, 255.5]): the 4:2:0 pooling truncates them itself, four
// at a time, with the saturating convert under round-toward-zero.
template <bool RAWC = false>
__device__ __forceinline__ void rgb_to_ycc(float r, float g, float b, float &y, float &cb, float &cr)
{
    y  = floorf((0.2990f * r + 0.5870f * g) + 0.1140f * b);          // `0 + x` is exact
    cb = __builtin_fmaf(0.5000f, b, (128.0f + -0.1687f * r) + -0.3313f * g);
    cr = (__builtin_fmaf(0.5000f, r, 128.0f) + -0.4187f * g) + -0.0813f * b;
    if constexpr (!RAWC) { cb = floorf(cb); cr = floorf(cr); }
}

// FDCT + quantise + zigzag scatter of one block held as 64 floats g[8y + x]; q / rq = modulated
// table (scale 8) and its correctly rounded reciprocal, in LDS, TRANSPOSED (q[8k + h]) so that the
// 8 entries one vertical pass needs are two 16-byte reads.  encode.swift:199-248.
//
// Quantiser: the reference computes RN(H / q) with a true division and rounds half away from
// zero.  Here  y0 = H * rq;  e = fma(-y0, q, H);  y1 = fma(e, rq, y0)  (Markstein's correction
// step) and  n = trunc(y1 + copysign(pred(0.5), y1)).  tools/verify_div.hip proves by exhaustion
// on the GPU that n equals the reference's integer for EVERY float numerator below 2^17 and
// EVERY divisor an 8-bit quantisation table can produce (7140 divisors x 1.2e9 numerators,
// profiles/r01_verify_div.txt); the fused kernel only runs for 8-bit formats.
__device__ __forceinline__ void fdct_quantise(const float (&g)[64], const float *q, const float *rq,
                                              uint32_t (&w)[32])
{
    float f[64];  // f[8k + y]: horizontal pass (encode.swift:193), level shift 2^(P-1) * 8
#pragma unroll
    for (int y = 0; y < 8; ++y) {
        float r[8], res[8];
#pragma unroll
        for (int x = 0; x < 8; ++x) r[x] = g[8 * y + x];
        fdct8<true>(r, 1024.0f, res);
#pragma unroll
        for (int k = 0; k < 8; ++k) f[8 * k + y] = res[k];
    }
    // Opaque empty asm statements pin the program order: without them LLVM re-orders the pure
    // arithmetic of all eight columns around the table reads and needs > 200 VGPRs (spills).
#pragma unroll
    for (int i = 0; i < 64; ++i) asm volatile("" : "+v"(f[i]));
    // w: 64 quantised coefficients, zigzag order, packed in pairs: a pair is converted and packed as soon as the later of its two
    // columns is done (zigzag neighbours lie at most one column apart, so few values wait)
    float zf[64];   // by zigzag index: y1 + copysign(pred(1/2), y1), whose truncation is the rounded coefficient
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        __builtin_amdgcn_sched_barrier(0);
        float r[8], res[8];
#pragma unroll
        for (int y = 0; y < 8; ++y) r[y] = f[8 * k + y];
        fdct8<false>(r, 0.0f, res);  // vertical pass (encode.swift:194)
#pragma unroll
        for (int h = 0; h < 8; ++h) {
            const float qq = q[8 * k + h], rr = rq[8 * k + h];
            const float y0 = res[h] * rr;
            const float e  = __builtin_fmaf(-y0, qq, res[h]);
            const float y1 = __builtin_fmaf(e, rr, y0);
            zf[zigzag_of(k, h)] = y1 + half_toward(y1);  // the conversion truncates; scatter (encode.swift:236-239)
        }
        pack_ready_pairs(zf, w, k, std::make_integer_sequence<int, 32>{});
    }
}

// Store the 64 blocks a wave has just quantised (one per lane, w = 128 bytes each).  A lane
// storing its own block would write 16 bytes of eight different 128-byte lines per instruction
// and revisit every line eight times; partial-line writes are what this kernel used to spend a
// third of its time on.  Instead the blocks go through a wave-private 8 KiB LDS buffer (chunk c
// of lane L at slot c ^ ((L >> 1) & 7), conflict-free for the stride-128-byte writes) and come
// back lane-linear: instruction i writes the complete lines of blocks 8i .. 8i + 7, each lane one
// 16-byte chunk, with the `nt` hint (streaming output).  `block` = index of the lane's block in
// `plane` (wave-uniform base; a plane of a 65535 x 65535 image exceeds 4 GiB, so the byte offset
// is formed in 64 bits by the storing lane), or ~0u for a block outside the plane; the producer's
// index reaches the storing lane through ds_bpermute.  LDS operations of one wave execute in order, so
// no barrier is needed; all 64 lanes must call this together.
__device__ __forceinline__ __attribute__((unused)) void wave_store_blocks(const uint32_t (&w)[32], uint32_t *stage, int lane,
                                                  int16_t *plane, uint32_t block)
{
    uint4 *mine = reinterpret_cast<uint4 *>(stage) + 8 * lane;
    const int sw = (lane >> 1) & 7;
#pragma unroll
    for (int c = 0; c < 8; ++c) mine[c ^ sw] = make_uint4(w[4 * c], w[4 * c + 1], w[4 * c + 2], w[4 * c + 3]);
    const uint4 *all = reinterpret_cast<const uint4 *>(stage);
    char *base = reinterpret_cast<char *>(plane);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int producer = 8 * i + (lane >> 3);
        const uint32_t blk = (uint32_t)__builtin_amdgcn_ds_bpermute(4 * producer, (int)block);
        const int c = (lane & 7) ^ ((producer >> 1) & 7);
        const uint4 v = all[64 * i + lane];
#ifdef JA_X_ENC_NOSTORE
        if (plane == nullptr)
#endif
        if (blk != ~0u) store_nt16(base + ((size_t)blk << 7) + 16 * c, v);
    }
}

// The same through a 4 KiB buffer, one half of the wave at a time (lanes 0..31 stage and the whole wave stores their
// 32 blocks, then lanes 32..63): half the LDS for eight more ds_write instructions.  For the 8-row-tile kernels,
// which want four workgroups on a CU.
__device__ __forceinline__ void wave_store_blocks_halves(const uint32_t (&w)[32], uint32_t *stage, int lane,
                                                         int16_t *plane, uint32_t block)
{
    const uint4 *all = reinterpret_cast<const uint4 *>(stage);
    char *base = reinterpret_cast<char *>(plane);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        if ((lane >> 5) == h) {
            uint4 *mine = reinterpret_cast<uint4 *>(stage) + 8 * (lane & 31);
            const int sw = (lane >> 1) & 7;
#pragma unroll
            for (int c = 0; c < 8; ++c) mine[c ^ sw] = make_uint4(w[4 * c], w[4 * c + 1], w[4 * c + 2], w[4 * c + 3]);
        }
        // the lanes read what the other half of the wave has just written: reconverge first (the compiler may otherwise place the
        // non-writers' reads in a branch of their own, ahead of the writers; k_generic_fused met exactly that)
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int producer = 32 * h + 8 * i + (lane >> 3);
            const uint32_t blk = (uint32_t)__builtin_amdgcn_ds_bpermute(4 * producer, (int)block);
            const int c = (lane & 7) ^ ((producer >> 1) & 7);
            const uint4 v = all[64 * i + lane];
#ifdef JA_X_ENC_NOSTORE
            if (plane == nullptr)
#endif
            if (blk != ~0u) store_nt16(base + ((size_t)blk << 7) + 16 * c, v);
        }
        __builtin_amdgcn_wave_barrier();   // the other half's writes stay behind these reads
    }
}

// SX, SY: chroma subsampling (1 or 2) per axis; RGB: input is RGB8 (else YCbCr8);
// CHROMA = false: single-plane image (only Y is produced);
// FASTIN: W % 8 == 0 and 8-byte aligned rows (vector loads for blocks inside the image).
// TY: luma block rows per tile.  16: a work-item transforms two luma blocks and (4:2:0) one chroma block.  8: one luma
// block, and half of the work-items a chroma block -- twice as many workgroups, for images whose 16-row tiles
// would not fill the chip (a 4096 x 4096 frame is 512 tiles of 16 rows: two waves per SIMD, each of them bound by
// its own instruction latency).
// POOLI: 4:2:0 only -- the 2 x 2 box filter in the integer domain (launches of several rounds; see POOL_INT below).
#ifdef JA_ENC_TIMELINE
}  // namespace
__device__ unsigned long long g_enc_timeline[2 * 32768];
namespace {
#endif
template <int SX, int SY, bool RGB, bool CHROMA, bool FASTIN, int TY = ETY, bool POOLI = false>
// 4:2:2 / 4:4:0 (chroma pooled per half tile) are built for TWO waves per SIMD: at three (168 VGPRs) the register allocator
// spills 9-19 registers of the FAST variants, and a spill reload waits with vmcnt(0) for every store in flight -- 4:4:0 at
// 4096 x 4096 35.3 -> 30.0 us, 4:2:2 33.4 -> 33.0 (profiles/r04_ab_encode_perhalf_two_waves.txt)
#ifndef JA_X_ENC_PERHALF_WAVES
#define JA_X_ENC_PERHALF_WAVES 2
#endif
__global__ __launch_bounds__(kThreads, (TY == 8 ? ((CHROMA && SX == 1 && SY == 1) ? 3 : 4) : (CHROMA && SX == 1 && SY == 1) ? 2 : (CHROMA && SX * SY == 2) ? JA_X_ENC_PERHALF_WAVES : 3)) void k_encode_fused(EncArgs a)
{
    // (waves per SIMD declared above: 4 for the 8-row tiles, 2 for 4:4:4 -- its 65 KiB of LDS and 256 VGPRs admit no
    // more -- 2 for 4:2:2 / 4:4:0, and 3 for the 16-row tiles of the JA_X_ENC_TY experiment)
    constexpr bool HALFSTAGE = TY == 8;                  // 4 KiB of store staging per wave instead of 8
    static_assert(TY == 16 || TY == 8, "tiles of 16 or 8 luma block rows");
    constexpr bool INTHREAD = SX == 1 && SY == 1;        // 4:4:4: chroma block == the luma block's pixels
    // 4:2:0: the 2 x 2 box filter in the integer domain (POOLI) removes 1.6 of the 50 instructions per pixel.  It pays where the
    // launch is several rounds of workgroups long (8192 x 8192: 88.6 against 91.5 us) and LOSES where the whole frame is one
    // round (4096 x 4096: 24.7 against 23.8 us -- every workgroup in the same phase, and the dependent convert / dot / permute
    // chain is longer than the float sum it replaces): the launcher picks (profiles/r04_ab_encode_integer_pooling.txt).
    static_assert(!POOLI || (CHROMA && SX == 2 && SY == 2), "integer pooling is the 4:2:0 box filter");
    constexpr bool POOL_INT = POOLI;
    // 4:2:2 / 4:4:0: the 8 luma block rows of one `half` already hold 256 chroma blocks (one per
    // work-item), so the chroma tile covers one half at a time and stays at 16 KiB
    constexpr bool PERHALF = CHROMA && SX * SY == 2 && TY == 16;   // (an 8-row tile of 4:2:2 / 4:4:0 holds one chroma block per work-item: a single pass)
    constexpr int CW = ETX * 8 / SX, CH = (PERHALF ? TY / 2 : TY) * 8 / SY;  // chroma samples per tile (or half)
    constexpr int CPITCH = CW / 4;                       // dwords per LDS row
    __shared__ uint32_t sc[(CHROMA && !INTHREAD) ? 2 * CH * CPITCH : 1];
    __shared__ __attribute__((aligned(16))) uint32_t stage_all[kThreads / 64][HALFSTAGE ? 32 * 32 : 64 * 32];  // 8 (4) KiB per wave
    // 4:4:4: the block's Cb / Cr samples wait here (packed 4 per dword, [dword][lane]) while the
    // luma block is transformed -- in registers they cost 32 VGPRs and the kernel spilled
    __shared__ uint32_t stash_all[(CHROMA && INTHREAD) ? kThreads / 64 : 1][(CHROMA && INTHREAD) ? 32 * 64 : 1];
    __shared__ float sq[3][64];   // modulated tables (scale 8) ...
    __shared__ float sr[3][64];   // ... and their correctly rounded reciprocals

    auto store_blocks = [](const uint32_t (&w)[32], uint32_t *stage, int lane, int16_t *plane, uint32_t block) {
        if constexpr (HALFSTAGE) wave_store_blocks_halves(w, stage, lane, plane, block);
        else wave_store_blocks(w, stage, lane, plane, block);
    };
#ifdef JA_ENC_TIMELINE   // development aid (tools/timeline_encode.py): start and end of every workgroup on the constant 100 MHz counter
    const unsigned long long tl_start = __builtin_amdgcn_s_memrealtime();
#endif
    const int img = blockIdx.y;
    // Wave priorities: everything up to the barrier in front of the chroma blocks runs at the top priority, the chroma blocks
    // one below.  Measured, not derived (profiles/r02_ab_encode_priority.txt): 4096 x 4096 4:2:0 23.1 against 25.0 us on the
    // same box; any split with the first phase above the default priority 0 gains 5-7 %.  (Layouts without that barrier gain
    // nothing from a raised priority: 4:4:4 loses 2 %.)
    if constexpr (CHROMA && !INTHREAD && !PERHALF) __builtin_amdgcn_s_setprio(3);
    const int tyi = blockIdx.x / a.tiles_x, txi = blockIdx.x - tyi * a.tiles_x;
    const int lbx = threadIdx.x & (ETX - 1), lby0 = threadIdx.x / ETX;
    const int lane = threadIdx.x & 63;
    uint32_t *stage = stage_all[threadIdx.x >> 6];
    uint32_t *stash = stash_all[(CHROMA && INTHREAD) ? threadIdx.x >> 6 : 0];

    // The tables are first needed by the luma FDCT.  Their quantum is REQUESTED here and turned into table entries only
    // after the tile's pixel loads have been issued, barrier included.  (Table first, barrier, then the pixel loads put two
    // memory latencies in a row at the head of every workgroup; a barrier later, in front of the FDCT, re-aligns the four
    // waves in the middle of their work: 4:2:0 23.5 -> 26.3 us.)
    // Measured (profiles/r02_ab_encode_tables_first.txt): -1.5 to -2 us for the 16-row tiles (4:2:2 30.3 -> 28.5, 4:4:4 37.0 ->
    // 35.0 us at 4096 x 4096), but +2.5 us for the 8-row tiles of grey and 4:2:0, whose single round of workgroups then
    // requests the whole frame in the same instant -- those keep the table in front (TABLES_LATE = false).
    constexpr bool TABLES_LATE = TY == 16;
    const bool qthread = threadIdx.x < 192 && (CHROMA || threadIdx.x < 64);
    const int qt = threadIdx.x >> 6, qk = threadIdx.x & 7, qh = (threadIdx.x >> 3) & 7;
    uint16_t qraw = 1;
    if (qthread) qraw = a.quanta[img * a.quanta_stride + 64 * a.qi[qt] + zigzag_of(qk, qh)];
    auto publish_tables = [&]() {
        if (qthread) {
            const float qv = modulate_entry(qk, qh, 8.0f, qraw);
            sq[qt][8 * qk + qh] = qv;                 // transposed: [k][h]
            sr[qt][8 * qk + qh] = 1.0f / qv;          // IEEE division: RN(1 / q)
        }
    };
    if constexpr (!TABLES_LATE) { publish_tables(); __syncthreads(); }

    // chroma blocks of the tile (or of one half of it) from the pooled LDS tile
    auto chroma_blocks = [&](int half) {
        constexpr int CBX = ETX / SX, CBY = (PERHALF ? TY / 2 : TY) / SY;  // chroma blocks per plane
#pragma unroll 1
        for (int c = threadIdx.x; c < 2 * CBX * CBY; c += kThreads) {
            const int pl = c / (CBX * CBY), r = c - pl * (CBX * CBY);
            const int cby = r / CBX, cbx = r - cby * CBX;
            const int bx = txi * CBX + cbx, by = tyi * (TY / SY) + half * CBY + cby;
            float g[64];
#pragma unroll
            for (int y = 0; y < 8; ++y) {
                const uint32_t *row = sc + (pl * CH + 8 * cby + y) * CPITCH + 2 * cbx;
                const uint32_t d0 = row[0], d1 = row[1];
                g[8 * y + 0] = ubyte<0>(d0); g[8 * y + 1] = ubyte<1>(d0);
                g[8 * y + 2] = ubyte<2>(d0); g[8 * y + 3] = ubyte<3>(d0);
                g[8 * y + 4] = ubyte<0>(d1); g[8 * y + 5] = ubyte<1>(d1);
                g[8 * y + 6] = ubyte<2>(d1); g[8 * y + 7] = ubyte<3>(d1);
            }
            uint32_t w[32];
#ifdef JA_X_ENC_NOCOMPUTE   // experiment (wrong coefficients): the kernel's memory traffic with next to no arithmetic
#pragma unroll
            for (int i = 0; i < 32; ++i) w[i] = __builtin_bit_cast(uint32_t, g[2 * i]) ^ __builtin_bit_cast(uint32_t, g[2 * i + 1]);
#else
            fdct_quantise(g, sq[1 + pl], sr[1 + pl], w);
#endif
            // 2 * CBX * CBY and CBX * CBY are multiples of 64: the loop is wave-uniform and a wave
            // never straddles the two planes
            const int plu = __builtin_amdgcn_readfirstlane(pl);
            const uint32_t off = (bx < a.ux[1 + plu] && by < a.uy[1 + plu]) ? (uint32_t)(by * a.ux[1 + plu] + bx) : ~0u;
            store_blocks(w, stage, lane, a.coef[1 + plu] + img * a.coef_stride[1 + plu], off);
        }
    };

    const uint8_t *base = a.px + img * a.px_stride;
#pragma unroll 1
    for (int half = 0; half < TY / 8; ++half) {
        const int lby = lby0 + 8 * half;
        const int bx = txi * ETX + lbx, by = tyi * TY + lby;
        float yv[64];

        // one pixel row of the block: Y stays in registers for the FDCT; Cb / Cr are pooled by
        // the box filter (every SY rows) into the LDS tile, or packed for the in-thread path
        float crow[2][2][8];  // [plane][row parity][x]
        auto emit_row = [&](int y, const float (&c0)[8], const float (&c1)[8], const float (&c2)[8]) {
#pragma unroll
            for (int x = 0; x < 8; ++x) {
                float yy, cb, cr;
                if constexpr (RGB) rgb_to_ycc<POOL_INT>(c0[x], c1[x], c2[x], yy, cb, cr);
                else { yy = c0[x]; cb = c1[x]; cr = c2[x]; }
                yv[8 * y + x] = yy;
                crow[0][y & 1][x] = cb;
                crow[1][y & 1][x] = cr;
            }
            if constexpr (CHROMA && INTHREAD) {
#pragma unroll
                for (int pl = 0; pl < 2; ++pl)
#pragma unroll
                    for (int d = 0; d < 2; ++d) {
                        uint32_t v = 0;
#pragma unroll
                        for (int i = 0; i < 4; ++i) v = __builtin_amdgcn_cvt_pk_u8_f32(crow[pl][y & 1][4 * d + i], i, v);
                        stash[(pl * 16 + 2 * y + d) * 64 + lane] = v;
                    }
            } else if constexpr (CHROMA) {
                if (SY == 1 || (y & 1)) {
                    constexpr float inv = 1.0f / (float)(SX * SY);
                    const int j = y / SY;
#pragma unroll
                    for (int pl = 0; pl < 2; ++pl) {
                        uint32_t packed[(8 / SX + 3) / 4] = {};
                        if constexpr (POOL_INT) {
                            // 4:2:0 (round 4): the window's four samples are truncated INTO ONE DWORD (v_cvt_pk_u8_f32 under
                            // round-toward-zero: floor for these positive values, encode.swift's per-pixel UInt8 conversion),
                            // summed by ONE v_dot4_u32_u8 with the weights 64, 64, 64, 64 -- 64 * sum < 2^16, so byte 1 of the
                            // result IS sum >> 2 = trunc(Float(sum) / 4) (encode.swift:419-421) -- and three v_perm_b32 gather
                            // the four results.  23 instructions per plane and pair of rows instead of 36.
                            float f[16];
#pragma unroll
                            for (int i = 0; i < 4; ++i) {
                                f[4 * i + 0] = crow[pl][0][2 * i]; f[4 * i + 1] = crow[pl][0][2 * i + 1];
                                f[4 * i + 2] = crow[pl][1][2 * i]; f[4 * i + 3] = crow[pl][1][2 * i + 1];
                            }
                            uint32_t q4[4], sum64[4];
                            trunc_pack16(f, q4);
#pragma unroll
                            for (int i = 0; i < 4; ++i) sum64[i] = __builtin_amdgcn_udot4(q4[i], 0x40404040u, 0u, false);
                            const uint32_t lo = __builtin_amdgcn_perm(sum64[1], sum64[0], 0x0c0c0501u);   // bytes: s0.b1, s1.b1, 0, 0
                            const uint32_t hi = __builtin_amdgcn_perm(sum64[3], sum64[2], 0x0c0c0501u);
                            packed[0] = __builtin_amdgcn_perm(hi, lo, 0x05040100u);
                        } else
#pragma unroll
                        for (int i = 0; i < 8 / SX; ++i) {
                            float sum;
                            if constexpr (SX == 2 && SY == 2)
                                sum = (crow[pl][0][2 * i] + crow[pl][0][2 * i + 1]) + (crow[pl][1][2 * i] + crow[pl][1][2 * i + 1]);
                            else if constexpr (SX == 2)
                                sum = crow[pl][y & 1][2 * i] + crow[pl][y & 1][2 * i + 1];
                            else
                                sum = crow[pl][0][i] + crow[pl][1][i];
                            // integer sum of n <= 4 bytes, exact; Float(sum) / n truncated (encode.swift:419-421).  The
                            // quotient is an integer plus 0, 1/n ... (n-1)/n: moved down by (1 - 1/n) / 2 it lies within
                            // 3/8 of that integer and never on a tie, so the convert's round-to-nearest IS the truncation
                            // (one exact FMA instead of a multiply and a floor)
                            packed[i >> 2] = __builtin_amdgcn_cvt_pk_u8_f32(__builtin_fmaf(sum, inv, -0.5f * (1.0f - inv)), i & 3, packed[i >> 2]);
                        }
                        uint32_t *row = sc + (pl * CH + (PERHALF ? lby0 : lby) * (8 / SY) + j) * CPITCH + lbx * (8 / SX) / 4;
#pragma unroll
                        for (int d = 0; d < (8 / SX + 3) / 4; ++d) row[d] = packed[d];
                    }
                }
            }
        };

        // ---- load 8x8 pixels as 8 rows of 24 bytes (edge replicate: encode.swift:415-417).
        //      All loads first (one latency), then one row at a time: left alone, the scheduler
        //      converts all 192 bytes to floats up front. ----
        uint32_t pix[8][6];
        const bool inside = 8 * bx + 8 <= a.W && 8 * by + 8 <= a.H;
        bool ragged = false;   // the block holds the image's last pixel column, or lies right of it
        int xlast = 7;
        if (FASTIN && inside) {
            // scalar row base (the tile's first pixel row + y rows: SALU) + one 32-bit per-lane offset for all eight rows:
            // no vector address arithmetic per row (it used to be three 64-bit multiply-adds per row)
#ifdef JA_X_ENC_L2LOAD   // experiment: every block reads the image's first tile (L2 hits)
            const uint8_t *tile0 = base;
#else
            const uint8_t *tile0 = base + ((size_t)(8 * tyi * TY) * a.W + (size_t)8 * txi * ETX) * 3;   // wave-uniform
#endif
            const uint32_t voff = ((uint32_t)(8 * lby) * (uint32_t)a.W + 8u * lbx) * 3u;   // < 64 rows x 65 535 px x 3 B
#pragma unroll
            for (int y = 0; y < 8; ++y) {
                const uint2 *row = reinterpret_cast<const uint2 *>(tile0 + (size_t)y * a.W * 3 + voff);
                const uint2 p0 = row[0], p1 = row[1], p2 = row[2];
                pix[y][0] = p0.x; pix[y][1] = p0.y; pix[y][2] = p1.x; pix[y][3] = p1.y; pix[y][4] = p2.x; pix[y][5] = p2.y;
            }
        } else {
            // Edge blocks, and every block of an image whose rows are not 8-byte aligned (any width): the block's 8 x 24 bytes
            // through a BUFFER RESOURCE (round 5; it was 192 clamped byte loads per block, and the variant spilled 36-48 VGPRs).
            // Row y of the block is image row min(8 by + y, H - 1) -- edge replicate downwards, encode.swift:417 -- and its 24
            // bytes start at pixel min(8 bx, W - 1): dword loads at any byte address (the hardware takes them), bytes past the end
            // of the image arrive as zeros (range check) and pixels right of column W - 1 are replaced below (`ragged`,
            // encode.swift:416).  The resource starts at the tile's first row inside the image, so that offsets stay 32-bit
            // for images beyond 4 GiB (<= 128 rows x 3 x 65 535 B).
            const int r0 = min(8 * tyi * TY, a.H - 1);
            const size_t left = (size_t)(a.H - r0) * a.W * 3;
            const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
                const_cast<uint8_t *>(base) + (size_t)r0 * a.W * 3, 0, (int)(uint32_t)(left < 0xffffffffull ? left : 0xffffffffull), 0x00020000);
            const int xs = min(8 * bx, a.W - 1);
            ragged = a.W - 1 - xs < 7;
            xlast = a.W - 1 - xs;
#pragma unroll
            for (int y = 0; y < 8; ++y) {
                const uint32_t off = ((uint32_t)(min(8 * by + y, a.H - 1) - r0) * (uint32_t)a.W + (uint32_t)xs) * 3u;
                const u32x4_t p0 = __builtin_amdgcn_raw_buffer_load_b128(rsrc, off, 0, 0);
                const u32x2_t p1 = __builtin_amdgcn_raw_buffer_load_b64(rsrc, off + 16, 0, 0);
                pix[y][0] = p0.x; pix[y][1] = p0.y; pix[y][2] = p0.z; pix[y][3] = p0.w; pix[y][4] = p1.x; pix[y][5] = p1.y;
                // The range check is per DWORD: where the image ends inside a dword of the 24 bytes (its last row, at the last
                // pixel: 3 (xlast + 1) bytes are left, not a multiple of 4) that dword arrives as 0 and its 1 .. 3 real bytes
                // are fetched one by one -- never a byte past the end of the caller's buffer.  One workgroup per image does this.
                const uint32_t n_in = left - off < 24 ? (uint32_t)(left - off) : 24u;   // (off < left: pixel (xs, row) exists)
                if (n_in < 24 && (n_in & 3)) {
                    const uint32_t k = n_in >> 2, at = off + 4 * k;
                    uint32_t v = __builtin_amdgcn_raw_buffer_load_b8(rsrc, at, 0, 0);
                    v |= (uint32_t)__builtin_amdgcn_raw_buffer_load_b8(rsrc, at + 1, 0, 0) << 8;   // (out of range: 0)
                    v |= (uint32_t)__builtin_amdgcn_raw_buffer_load_b8(rsrc, at + 2, 0, 0) << 16;
#pragma unroll
                    for (int d = 0; d < 6; ++d) pix[y][d] = (uint32_t)d == k ? v : pix[y][d];
                }
            }
        }
        if (TABLES_LATE && half == 0) { publish_tables(); __syncthreads(); }   // the pixel loads are in flight: the barrier waits under them
#pragma unroll
        for (int y = 0; y < 8; ++y) {
            __builtin_amdgcn_sched_barrier(0);
            float c[3][8];
#pragma unroll
            for (int x = 0; x < 8; ++x)
#pragma unroll
                for (int ch = 0; ch < 3; ++ch) {
                    const int byte = 3 * x + ch;
                    const uint32_t dw = pix[y][byte >> 2];
                    c[ch][x] = (byte & 3) == 0 ? ubyte<0>(dw) : (byte & 3) == 1 ? ubyte<1>(dw)
                               : (byte & 3) == 2 ? ubyte<2>(dw) : ubyte<3>(dw);
                }
            if (ragged) {   // edge replicate to the right: pixel x > W - 1 is pixel W - 1 (encode.swift:416)
#pragma unroll
                for (int x = 1; x < 8; ++x)
#pragma unroll
                    for (int ch = 0; ch < 3; ++ch) c[ch][x] = x <= xlast ? c[ch][x] : c[ch][x - 1];
            }
#ifdef JA_X_ENC_NOCOMPUTE
#pragma unroll
            for (int x = 0; x < 8; ++x) yv[8 * y + x] = c[0][x] + c[1][x] + c[2][x];
            if (y == 7 && CHROMA && !INTHREAD) {
                uint32_t *row = sc + (lby * 4 * CPITCH + lbx) % (2 * CH * CPITCH);
                row[0] = pix[0][0];
            }
#else
            emit_row(y, c[0], c[1], c[2]);
#endif
#pragma unroll
            for (int x = 0; x < 8; ++x) asm volatile("" : "+v"(yv[8 * y + x]));
        }

        // ---- luma (and 4:4:4 chroma) blocks of this position ----
        __builtin_amdgcn_sched_barrier(0);
        {
            uint32_t w[32];
#ifdef JA_X_ENC_NOCOMPUTE
#pragma unroll
            for (int i = 0; i < 32; ++i) w[i] = __builtin_bit_cast(uint32_t, yv[2 * i]) ^ __builtin_bit_cast(uint32_t, yv[2 * i + 1]);
#else
            fdct_quantise(yv, sq[0], sr[0], w);
#endif
            const uint32_t off = (bx < a.ux[0] && by < a.uy[0]) ? (uint32_t)(by * a.ux[0] + bx) : ~0u;
            store_blocks(w, stage, lane, a.coef[0] + img * a.coef_stride[0], off);
        }
        if constexpr (CHROMA && INTHREAD) {
#pragma unroll 1
            for (int pl = 0; pl < 2; ++pl) {
                float g[64];
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const uint32_t v = stash[(pl * 16 + i) * 64 + lane];
                    g[4 * i + 0] = ubyte<0>(v); g[4 * i + 1] = ubyte<1>(v);
                    g[4 * i + 2] = ubyte<2>(v); g[4 * i + 3] = ubyte<3>(v);
                }
                uint32_t w[32];
                fdct_quantise(g, sq[1 + pl], sr[1 + pl], w);
                const uint32_t off = (bx < a.ux[1 + pl] && by < a.uy[1 + pl]) ? (uint32_t)(by * a.ux[1 + pl] + bx) : ~0u;
                store_blocks(w, stage, lane, a.coef[1 + pl] + img * a.coef_stride[1 + pl], off);
            }
        }
        if constexpr (PERHALF) {
            __syncthreads();          // the half's chroma samples are pooled
            chroma_blocks(half);
            if (half == 0) __syncthreads();   // ... and consumed before the next half overwrites them
        }
    }
    if constexpr (CHROMA && !INTHREAD && !PERHALF) {
        __syncthreads();
        __builtin_amdgcn_s_setprio(2);   // the tile's tail; the waves that are still converting and transforming luma go first
#ifdef JA_X_ENC_NOCHROMA   // experiment (no chroma coefficients): what does the two-wave chroma tail of a tile cost?
        if (a.W < 0)
#endif
        chroma_blocks(0);
    }
#ifdef JA_ENC_TIMELINE
    {
        const unsigned wg = blockIdx.y * gridDim.x + blockIdx.x;
        if (threadIdx.x == 0 && wg < 32768) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            g_enc_timeline[2 * wg] = tl_start; g_enc_timeline[2 * wg + 1] = __builtin_amdgcn_s_memrealtime();
        }
    }
#endif
}

}  // namespace
#ifdef JA_ENC_TIMELINE
extern "C" int jpeg_amd_debug_encode_timeline(unsigned long long *h_out, size_t n)
{
    return (int)hipMemcpyFromSymbol(h_out, HIP_SYMBOL(jpeg_amd::g_enc_timeline), n * sizeof(unsigned long long));
}
#endif

// development switch (-DJA_X_ENC_TY=8 / 16, tools/build_exp.sh): forces the tile height of the grey / 4:2:0 encode kernels
static int encode_ty_override()
{
#ifdef JA_X_ENC_TY
    return JA_X_ENC_TY;
#else
    return 0;
#endif
}

bool fused_encode_supported(const jpeg_amd_layout &L)
{
    if (L.precision != 8) return false;
    auto units = [](int size, int stride) { return size / stride + (size % stride != 0 ? 1 : 0); };
    for (int p = 0; p < L.nplanes; ++p) {
        if (L.units_x[p] != units(L.width * L.factor_x[p], 8 * L.scale_x)) return false;
        if (L.units_y[p] != units(L.height * L.factor_y[p], 8 * L.scale_y)) return false;
    }
    if (L.factor_x[0] != L.scale_x || L.factor_y[0] != L.scale_y) return false;  // luma at full resolution
    if (L.nplanes == 1) return true;
    if (L.nplanes != 3) return false;
    if (L.scale_x > 2 || L.scale_y > 2) return false;
    for (int p = 1; p < 3; ++p)
        if (L.factor_x[p] != 1 || L.factor_y[p] != 1) return false;
    return true;
}

hipError_t launch_fused_encode(hipStream_t stream, int n_images, const jpeg_amd_layout &L,
                               const uint8_t *d_pixels, size_t pixel_stride, bool rgb, QuantaRef q,
                               const PlaneSetMut &coef)
{
    EncArgs a{};
    a.px = d_pixels; a.px_stride = pixel_stride; a.W = L.width; a.H = L.height;
    a.quanta = q.d_quanta; a.quanta_stride = q.image_stride;
    for (int p = 0; p < 3; ++p) {
        const int s = p < L.nplanes ? p : 0;
        a.coef[p] = static_cast<int16_t *>(coef.ptr[s]);
        a.coef_stride[p] = coef.stride[s];
        a.ux[p] = p < L.nplanes ? L.units_x[p] : 0;
        a.uy[p] = p < L.nplanes ? L.units_y[p] : 0;
        a.qi[p] = L.qi[s];
    }
    const bool chroma = L.nplanes == 3;
    const int sx = chroma ? L.scale_x : 1, sy = chroma ? L.scale_y : 1;
    // tiles must cover the luma blocks AND the pixels under every chroma block
    const int need_x = chroma ? max(a.ux[0], sx * a.ux[1]) : a.ux[0];
    const int need_y = chroma ? max(a.uy[0], sy * a.uy[1]) : a.uy[0];
    a.tiles_x = (need_x + ETX - 1) / ETX;
    // Every layout takes 8-row tiles (one luma block per work-item; 99-124 VGPRs: four waves per SIMD, three for 4:4:4 whose
    // parked chroma samples cost 32 KiB of LDS).  Grey and 4:2:0 since round 2 (4096 x 4096 4:2:0 30.9 -> 28.8 us, 2048 x 2048 23.3 ->
    // 15.4); 4:2:2 / 4:4:0 / 4:4:4 since round 5 -- an 8-row tile of 4:2:2 / 4:4:0 holds exactly one chroma block per work-item, so the
    // per-half chroma tiles of the 16-row kernels (two waves per SIMD, 151-183 VGPRs) are not needed: 8192 x 8192 4:2:2 112 -> 104 us,
    // 4:4:4 155 -> 128, 4:4:0 118 -> 98; 4096 x 4096 33.3 -> 28.0, 40.2 -> 39.2, 32.5 -> 27.5 (profiles/r05_ab_encode_8_row_tiles.txt).
    // The 16-row instantiations are built for the JA_X_ENC_TY experiment alone.
    const int ty = encode_ty_override() == 16 ? 16 : 8;
    const int tiles_y = (need_y + ty - 1) / ty;
    if (a.tiles_x * tiles_y == 0 || n_images == 0) return hipSuccess;
    const dim3 grid(a.tiles_x * tiles_y, n_images);
    // more workgroups than are resident at once (four 8-row tiles per CU)?  Then the launch is several rounds long.
    // (what is resident at once: the occupancy of the instantiation in question x CUs, cached per device)
    const bool several_rounds = (long)a.tiles_x * tiles_y * n_images > (long)resident_workgroups_of<k_encode_fused<2, 2, true, true, true, 8, false>>(4);
    // (the vector-load path addresses a tile's rows with 32-bit byte offsets: 16 block rows x 8 x W x 3 B must stay below 2^32)
    const bool fast = (L.width & 7) == 0 && (pixel_stride & 7) == 0 && (reinterpret_cast<uintptr_t>(d_pixels) & 7) == 0 &&
                      L.width <= (1 << 23);
#define JA_E8(SX_, SY_, RGB_, CH_, F_) hipLaunchKernelGGL((k_encode_fused<SX_, SY_, RGB_, CH_, F_, 8>), grid, dim3(kThreads), 0, stream, a)
#define JA_E8I(SX_, SY_, RGB_, CH_, F_) hipLaunchKernelGGL((k_encode_fused<SX_, SY_, RGB_, CH_, F_, 8, true>), grid, dim3(kThreads), 0, stream, a)
#ifdef JA_X_ENC_TY
#define JA_ET(SX_, SY_, RGB_, CH_, F_) do { if (ty == 8) JA_E8(SX_, SY_, RGB_, CH_, F_); else hipLaunchKernelGGL((k_encode_fused<SX_, SY_, RGB_, CH_, F_>), grid, dim3(kThreads), 0, stream, a); } while (0)
#else
#define JA_ET(SX_, SY_, RGB_, CH_, F_) JA_E8(SX_, SY_, RGB_, CH_, F_)
#endif
#define JA_E2(RGB_, F_)                                         \
    do {                                                        \
        if (!chroma) JA_ET(1, 1, RGB_, false, F_);              \
        else if (sx == 2 && sy == 2 && ty == 8 && several_rounds) JA_E8I(2, 2, RGB_, true, F_); \
        else if (sx == 2 && sy == 2) JA_ET(2, 2, RGB_, true, F_); \
        else if (sx == 2 && sy == 1) JA_ET(2, 1, RGB_, true, F_); \
        else if (sx == 1 && sy == 2) JA_ET(1, 2, RGB_, true, F_); \
        else JA_ET(1, 1, RGB_, true, F_);                       \
    } while (0)
    if (rgb) { if (fast) JA_E2(true, true); else JA_E2(true, false); }
    else     { if (fast) JA_E2(false, true); else JA_E2(false, false); }
#undef JA_E2
#undef JA_ET
#undef JA_E8I
#undef JA_E8
    return hipGetLastError();
}

}  // namespace jpeg_amd
