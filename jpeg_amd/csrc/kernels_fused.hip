// kernels_fused.hip -- fused Spectral -> pixels fast path for the built-in 8-bit formats.
//
// Replaces idct() -> interleaved(cosite: false) -> unpack(as:) (decode.swift:4154, 4182,
// 4294) for ycc8 images whose luma has the full sampling factor and whose chroma planes are
// subsampled 1x or 2x per axis (4:4:4, 4:2:2, 4:4:0, 4:2:0), and for y8 images, without
// materialising Planar / Rectangular in HBM.  Two launches (one for y8, 4:4:4 and 4:2:2, where
// the work-items transform the Cb and Cr blocks under their strip themselves):
//
//   k_chroma_idct   Cb and Cr: dequantise + IDCT, clamp, store as uint8 planes (a scratch of
//                   0.5 B/px for 4:2:0 -- small enough to stay in L2 / Infinity Cache).
//   k_luma_fused    one luma 8x8 block per work-item: dequantise + IDCT in registers; the
//                   workgroup stages its chroma tile (+1 sample halo, clamped to the padded
//                   plane like decode.swift:4245-4246) in LDS; bilinear upsample, YCbCr->RGB,
//                   pack and store 8 rows x 24 B.
//
// Exactness of the upsample shortcut.  For centred 2x upsampling the reference's weights are
// t in {1/4, 3/4} (decode.swift:4231-4251), so u00*(1-t) + u01*t etc. are sums of small
// integers times binary fractions: every intermediate is exactly representable in binary32
// (<= 12 significant bits).  (3a + b) / 4 evaluated with an FMA is therefore the SAME value,
// and round-half-away of an exact multiple of 1/16 is floor(v + 0.5).  The index clamp
// max(i, 0) at the left/top edge reproduces the reference's t = 0 case (decode.swift:4240,
// 4250) because 0.25*c + 0.75*c == c exactly.  Everything that rounds (dequantise, IDCT,
// colour matrix) is evaluated op-for-op as in dct.hpp / the reference.
//
// Development switches (never defined in the product build; tools/build_exp.sh makes A/B builds,
// DESIGN.md section 6 quotes the measurements): JA_PHASE_PROFILE (per-phase cycle counters,
// tools/phase_profile.py), JA_X_NOIDCT / JA_X_NOCOLOR / JA_X_NOSTORE / JA_X_NOCTILE (the kernel without its transform /
// without its upsampling and colour arithmetic / without its stores / without the chroma tile copy: tools/ablate.sh),
// JA_X_SKIPK1 / JA_X_SKIPK2 (one launch of the pair only, tools/probe_overlap.py),
// JA_X_NO_IN420 / JA_X_FORCE_IN420 (4:2:0 chroma in the strip walk never / whenever the strips are wide),
// JA_X_IN420_NT (its neighbour fetches with the `nt` hint: 8 % slower, they are re-used out of L2).
#pragma clang fp contract(off)

#include "dct.hpp"
#include "kernels.hpp"
#include "upsample.hpp"

#include <algorithm>
#include <cstdlib>

namespace jpeg_amd {

namespace {

constexpr int kThreads = 256;
// a strip is BX x BY luma blocks, BX * BY == 64 (one block per work-item): 32 x 2, or 16 x 4 for
// images whose width leaves the last 32-block strip half empty (1920 px = 7.5 strips of 32)

// ---------------------------------------------------------------------------------------
// K1: chroma planes -> uint8 samples.  blockIdx.z selects the plane (same geometry).
// ---------------------------------------------------------------------------------------
struct ChromaArgs {
    const int16_t *coef[2];
    size_t coef_stride[2];
    uint8_t *out[2];
    size_t out_stride;
    const uint16_t *quanta;
    size_t quanta_stride;
    int qi[2];
    int ux, first_block, end_block;   // this launch transforms blocks [first_block, end_block) of every image
};

__global__ __launch_bounds__(kThreads) void k_chroma_idct(ChromaArgs a)
{
    __shared__ float sq[64];
    const int img = blockIdx.y, pl = blockIdx.z;
    // the table's quantum is requested first, the block's coefficients right behind it: one memory latency at the head of
    // the workgroup instead of two (table, barrier, then the coefficient loads)
    const int qk = threadIdx.x & 7, qh = (threadIdx.x >> 3) & 7;
    uint16_t qraw = 1;
    if (threadIdx.x < 64) qraw = a.quanta[img * a.quanta_stride + 64 * a.qi[pl] + zigzag_of(qk, qh)];
    const int b = a.first_block + blockIdx.x * kThreads + threadIdx.x;
    const bool mine = b < a.end_block;
    const int bc = mine ? b : a.end_block - 1;   // work-items past the range re-read the last block and store nothing
    const int by = bc / a.ux, bx = bc - by * a.ux;

    const uint4 *src = reinterpret_cast<const uint4 *>(a.coef[pl] + img * a.coef_stride[pl] + (size_t)64 * bc);
    uint32_t w[32];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const uint4 v = src[i];
        w[4 * i + 0] = v.x; w[4 * i + 1] = v.y; w[4 * i + 2] = v.z; w[4 * i + 3] = v.w;
    }
    if (threadIdx.x < 64) sq[threadIdx.x] = modulate_entry(qk, qh, 0.125f, qraw);
    __syncthreads();
    if (!mine) return;
    float g[64];
    idct_block(w, sq, 128.5f, g);  // level = 2^(P-1) + 0.5, P = 8

    const size_t pitch = (size_t)8 * a.ux;
    uint8_t *dst = a.out[pl] + img * a.out_stride + (size_t)8 * by * pitch + 8 * bx;
#pragma unroll
    for (int y = 0; y < 8; ++y) {
        // clamp [0, 255] + truncate == saturating convert of floor(v)
        uint32_t lo = 0, hi = 0;
#pragma unroll
        for (int x = 0; x < 4; ++x) {
            lo = __builtin_amdgcn_cvt_pk_u8_f32(floorf(g[8 * y + x]), x, lo);
            hi = __builtin_amdgcn_cvt_pk_u8_f32(floorf(g[8 * y + 4 + x]), x, hi);
        }
        *reinterpret_cast<uint2 *>(dst + y * pitch) = make_uint2(lo, hi);
    }
}

// The same, persistent and software-pipelined: a wave walks units of 64 consecutive blocks (u, u + nwaves, ...) and the
// 128 bytes of its NEXT block are requested before the current one is transformed.  k_chroma_idct starts all its
// workgroups at once -- the whole chip loads, then the whole chip computes, then it stores (a round is memory time PLUS
// arithmetic time: 24 us for 100 MB and 7 M VALU instructions at 8192 x 8192); here every wave always has a block in
// flight while it works on another.  Units are per (image, plane): `units_per_plane` x 64 blocks cover the launch's
// block range, so a wave never straddles two tables.
struct ChromaPersistArgs {
    ChromaArgs c;
    int n_images, units_per_plane, total_units;
};

__global__ __launch_bounds__(kThreads, 3) void k_chroma_idct_persist(ChromaPersistArgs p)
{
    const ChromaArgs &a = p.c;
    __shared__ float sqw[kThreads / 64][64];
    const int lane0 = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float *sq = sqw[wave];
    const int nwaves = gridDim.x * (kThreads / 64);
    int u = blockIdx.x * (kThreads / 64) + wave;
    if (u >= p.total_units) return;

    auto source = [&](int unit, int ln, int &img, int &pl, int &b) -> const uint4 * {
        const int ip = unit / p.units_per_plane;
        img = ip >> 1; pl = ip & 1;
        b = a.first_block + 64 * (unit - ip * p.units_per_plane) + ln;
        const int bc = min(b, a.end_block - 1);       // lanes past the range re-read the last block; nothing is stored
        return reinterpret_cast<const uint4 *>(a.coef[pl] + img * a.coef_stride[pl] + (size_t)64 * bc);
    };
    uint32_t wn[32];
    auto fetch = [&](const uint4 *src) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const uint4 v = src[i];
            wn[4 * i + 0] = v.x; wn[4 * i + 1] = v.y; wn[4 * i + 2] = v.z; wn[4 * i + 3] = v.w;
        }
    };
    int img, pl, b;
    fetch(source(u, lane0, img, pl, b));
    int table_of = -1;
    for (; u < p.total_units; u += nwaves) {
        int lane = lane0;
        asm volatile("" : "+v"(lane));
        (void)source(u, lane, img, pl, b);
        const int ip = 2 * img + pl;
        if (ip != table_of) {   // wave-uniform
            const int k = lane & 7, h = lane >> 3;
            sq[lane] = modulate_entry(k, h, 0.125f, a.quanta[img * a.quanta_stride + 64 * a.qi[pl] + zigzag_of(k, h)]);
            table_of = ip;
        }
        uint32_t w[32];
#pragma unroll
        for (int i = 0; i < 32; ++i) w[i] = wn[i];
        if (u + nwaves < p.total_units) {
            int i2, p2, b2;
            fetch(source(u + nwaves, lane, i2, p2, b2));
        }
        float g[64];
        idct_block(w, sq, 128.5f, g);  // level = 2^(P-1) + 0.5, P = 8
        asm volatile("" : "+v"(lane));
        if (b < a.end_block) {
            const int by = b / a.ux, bx = b - by * a.ux;
            const size_t pitch = (size_t)8 * a.ux;
            uint8_t *dst = a.out[pl] + img * a.out_stride + (size_t)8 * by * pitch + 8 * bx;
#pragma unroll
            for (int y = 0; y < 8; ++y) {
                // clamp [0, 255] + truncate == saturating convert of floor(v)
                uint32_t lo = 0, hi = 0;
#pragma unroll
                for (int x = 0; x < 4; ++x) {
                    lo = __builtin_amdgcn_cvt_pk_u8_f32(floorf(g[8 * y + x]), x, lo);
                    hi = __builtin_amdgcn_cvt_pk_u8_f32(floorf(g[8 * y + 4 + x]), x, hi);
                }
                *reinterpret_cast<uint2 *>(dst + y * pitch) = make_uint2(lo, hi);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------
// K2: luma IDCT + chroma upsample + colour + store
// ---------------------------------------------------------------------------------------
// Division by a launch-invariant: q = mulhi(n, floor((2^32 - 1) / d)) is the quotient or one below it for every n < 2^32, one
// compare fixes it.  The strip walks divide a dozen times per trip (strip -> image, row, column; strips left); as true
// divisions that is ~400 mostly scalar, serial instructions at the head of every trip.
struct FastDiv {
    uint32_t d, m;
    __device__ __forceinline__ void set(uint32_t div) { d = div; m = 0xffffffffu / div; }
    __device__ __forceinline__ uint32_t div(uint32_t n, uint32_t &r) const
    {
        uint32_t q = __umulhi(n, m);
        r = n - q * d;
        if (r >= d) { ++q; r -= d; }
        return q;
    }
};

struct LumaArgs {
    const int16_t *coef;
    size_t coef_stride;
    const uint8_t *cb, *cr;  // uint8 planes [ph_c][pw_c] (unused for grey and for 4:4:4)
    size_t c_stride;
    const int16_t *ccoef[2]; // 4:4:4: the chroma coefficient planes themselves (same geometry as luma)
    size_t ccoef_stride[2];
    int cqi[2];
    const uint16_t *quanta;
    size_t quanta_stride;
    int qi;
    int ux, uy;              // luma units
    int pw_c, ph_c;          // padded chroma plane size
    int W, H;
    uint8_t *out;
    size_t out_stride;
    int tiles_x, tiles_per_image;
    int first_tile, total_tiles;   // this launch walks strips [first_tile, total_tiles) of the call
    // Batches: n_images > 0 gives image i to the workgroups with blockIdx.x % 8 == i % 8, i.e. to ONE XCD (workgroups
    // are dealt to the eight XCDs round-robin; tools/probe_xcd.hip).  The halo rows and columns of a strip's chroma tile
    // are the neighbouring strips' own samples: with all strips of an image on one XCD they are found in its L2 instead
    // of being fetched again by up to eight of them (a 1080p batch fetches its chroma planes 3.3 times over).  The
    // partition is a function of blockIdx.x alone: correct whatever the hardware does with it.  Opt-in (see
    // xcd_images_enabled: fewer fetches, no faster).
    int xcd_images;
    int quad;                // 4:2:0 in one launch, four stacked strips per workgroup sharing a chroma tile (k_luma_fused QUAD)
};

// SX, SY: chroma subsampling per axis (1 or 2); MODE: 0 = YCbCr bytes, 1 = RGB bytes;
// CHROMA = false: single-plane (grey) image.
// FAST: W % 16 == 0 and 16-byte aligned rows, so every 16-byte chunk of a row segment is either
// entirely inside the image or entirely outside (no byte-wise tail code in the hot path).
//
// Persistent, fully independent WAVES.  The unit of work is a strip of BX x BY luma blocks
// (32 x 2 = 256 x 16 px, or 16 x 4 = 128 x 32 px); wave g of the launch walks strips g,
// g + nwaves, ...  Lane l is block (l % BX, l / BX) of the strip (row-major).  Everything a wave touches in LDS is private to it: there is no workgroup barrier,
// waves drift apart and their memory and arithmetic phases interleave on the SIMD.
//
// What bounds this kernel is HBM at the rate the chip sustains for a 1 : 1 read / write stream (5.4-6.1 TB/s,
// tools/probe_stream.hip; the kernel moves its bytes at 5.6); the arithmetic has to stay hidden
// behind that with three waves per SIMD (one wave alone issues a VALU instruction only every ~7-9 cycles,
// tools/probe_mix.hip).  So the design keeps waves from parking:
//   - the 8 KiB of coefficients of the NEXT strip are fetched by LDS-DMA
//     (global_load_lds_dwordx4) into the wave's LDS buffer while it works on the current one;
//     no VGPRs are spent on the prefetch.  The LDS image is lane-linear (a DMA requirement);
//     an XOR swizzle on the global SOURCE address makes the later per-work-item ds_read_b128
//     (stride 128 B) bank-conflict-free;
//   - the chroma samples under the strip are requested before IDCT pass 1 and land during it;
//   - no register spills (a spill reload waits with vmcnt(0) and thereby for every store in
//     flight): the 64 luma samples are packed into 16 VGPRs after the IDCT, phases are kept
//     apart with scheduling barriers, strip geometry lives in SGPRs.
// Pixels leave through an LDS staging row so that every global store instruction writes whole
// 16-byte chunks of contiguous 768-byte row segments.
// One 16-byte-per-lane LDS-DMA: lane l's 16 B at `g` land at LDS byte address lds + 16 l.
// Issued from inline asm on purpose: hipcc cannot tell which LDS array a DMA targets, so
// after a __builtin_amdgcn_global_load_lds it makes the NEXT LDS read of any array wait with
// vmcnt(0) -- i.e. for the prefetch it was supposed to overlap.  Hidden from the compiler, the
// DMA is only waited for by the explicit s_waitcnt at the top of the next strip.  (Extra
// outstanding VM operations can only make the compiler's own counted waits longer, never
// shorter, because loads retire in order.)
__device__ __forceinline__ void lds_dma16(const void *g, uint32_t lds)
{
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off nt" ::"v"(g), "s"(lds) : "memory");
}
// Same with a scalar 64-bit base + per-lane 32-bit byte offset (no vector address arithmetic).
__device__ __forceinline__ void lds_dma16_s(uint64_t sbase, uint32_t voff, uint32_t lds)
{
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %0 nt" ::"s"(sbase), "v"(voff), "s"(lds) : "memory");
}
// The same two without the `nt` hint: for coefficients that neighbouring strips fetch again soon
// (the chroma blocks around a 4:2:0 strip) and should therefore stay in L2.
#ifdef JA_X_IN420_NT
#define JA_KEEP_HINT " nt"
#else
#define JA_KEEP_HINT ""
#endif
__device__ __forceinline__ void lds_dma16_keep(const void *g, uint32_t lds)
{
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" JA_KEEP_HINT ::"v"(g), "s"(lds) : "memory");
}
__device__ __forceinline__ void lds_dma16_s_keep(uint64_t sbase, uint32_t voff, uint32_t lds)
{
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %0" JA_KEEP_HINT ::"s"(sbase), "v"(voff), "s"(lds) : "memory");
}
// 4 bytes per lane: lane l's dword lands at LDS byte address lds + 4 l.
__device__ __forceinline__ void lds_dma4_s(uint64_t sbase, uint32_t voff, uint32_t lds)
{
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, %0" ::"s"(sbase), "v"(voff), "s"(lds) : "memory");
}

// Split arrive / wait on an LDS counter (QUAD).  The LDS executes one wave's operations in issue order, so a ds_add issued
// behind the wave's tile writes (or its last tile reads) is performed behind them: no s_waitcnt at the arrive.  The waiting
// side polls with plain LDS reads; what it reads from the tile after the poll has succeeded is issued, and therefore
// performed, after the read that saw the counter.  Relaxed atomics + compiler barriers on purpose: a release / acquire at
// workgroup scope would make hipcc wait with vmcnt(0) -- for the coefficient DMA in flight and for the pixel stores.
__device__ __forceinline__ void lds_arrive(uint32_t *counter, int lane)
{
    asm volatile("" ::: "memory");
    if (lane == 0) (void)__hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    asm volatile("" ::: "memory");
}
__device__ __forceinline__ void lds_wait_ge(uint32_t *counter, uint32_t target)
{
#ifdef JA_X_NOSYNC   // experiment (wrong pixels): what do the waits of the QUAD walk cost?
    return;
#endif
    asm volatile("" ::: "memory");
    while ((uint32_t)__builtin_amdgcn_readfirstlane((int)__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) < target)
        __builtin_amdgcn_s_sleep(1);
    asm volatile("" ::: "memory");
}

#ifdef JA_PHASE_PROFILE
// development aid (tools/phase_profile.py): wall cycles each wave spends per phase of a strip
__device__ unsigned long long g_phase_cycles[4096 * 16];
__device__ unsigned long long g_wave_info[4096 * 4];   // start tick, end tick (100 MHz counter), HW_ID, XCC_ID
#define JA_PHASE(i)                                                       \
    {                                                                     \
        const unsigned long long t_now = __builtin_readcyclecounter();    \
        phase_acc[i] += t_now - t_prev;                                   \
        t_prev = t_now;                                                   \
    }
#else
#define JA_PHASE(i)
#endif

// Waves per SIMD a variant is built for: what its LDS footprint admits (three workgroups of ~51 KiB per CU for 4:2:0 and
// grey; the layouts with a full-width or full-height chroma tile and 4:4:4 need 58-75 KiB per workgroup: two).
template <int SX, int SY, bool CHROMA, bool DIRECT, bool ALIAS = false>
constexpr int luma_waves_per_simd() { return ALIAS ? 4 : (CHROMA && (SX == 1 || SY == 1)) ? 2 : 3; }

template <int SX, int SY, int MODE, bool CHROMA, bool FAST, int BX, bool STRIP420 = false, bool DIRECT = false, bool ALIAS = false, bool QUAD_ = false>
__global__ __launch_bounds__(kThreads, (luma_waves_per_simd<SX, SY, CHROMA, DIRECT, ALIAS>())) void k_luma_fused(LumaArgs a)
{
    // QUAD (4:2:0 in one launch, opt-in JPEG_AMD_QUAD=1): the four waves of a workgroup take four vertically stacked
    // strips and SHARE one chroma tile in LDS.  Each wave transforms, in ONE pass, the 32 chroma blocks under its own strip,
    // a quarter of the 64 blocks that supply the sample rows above and below the stack, and the side blocks of its row
    // (56 work-items busy); after a barrier every wave finds its halo rows in its neighbours' samples.  2.0 IDCT passes
    // per strip where STRIP420 needs 2.7 (44 + 64 work-items, the second pass at two thirds of the arithmetic) and the
    // two launches 1.5 -- and no chroma samples through HBM.
    // The same for 16 x 4 strips (128 x 32 pixels, what 1920 x 1080 is made of): a strip has two chroma block rows, so TWO
    // stacked strips fill a wave's pass (32 own blocks, 16 of the 32 halo-row blocks, 8 side blocks, 4 corner blocks: 60
    // work-items) and a workgroup walks two independent stacks that share nothing but the barrier.
    constexpr bool QUAD = QUAD_ && STRIP420 && CHROMA && SX == 2 && SY == 2;
    constexpr int QS = BX == 32 ? 4 : 2;                 // QUAD: strips (= waves) per stack
    constexpr int QG = (kThreads / 64) / QS;             // QUAD: stacks per workgroup
    // ALIAS (two-launch 4:2:0): the chroma tile lives INSIDE the wave's coefficient buffer.  The buffer is only needed
    // from the prefetch of the next strip's coefficients to their read-back at the top of that strip, the tile from
    // there to the last chroma read of the pixel rows (row 5) -- so the prefetch is issued after row 5 instead of
    // after the transform and the two never overlap.  The wave's LDS drops from 12.9 to 9.75 KiB: FOUR waves fit a
    // SIMD (the kernel needs 125 VGPRs), and 4 096 resident waves take the 16 384 strips of an 8192 x 8192 image in
    // exactly four rounds instead of 5.33 (the thin sixth round of section 6.1 is gone).
    static_assert(!ALIAS || (CHROMA && SX == 2 && SY == 2 && !STRIP420 && !DIRECT), "ALIAS is built for the two-launch 4:2:0 path");
    // DIRECT: no LDS coefficient buffer and no LDS-DMA -- a work-item loads its own block (8 x 16 B of its 128-byte
    // line) straight into registers, one strip AHEAD: the loads for the next strip are issued in the middle of the
    // pixel rows, when half of the luma samples are consumed and their registers are free.  Saves the 8 DMA
    // instructions per strip (60-185 cycles each to issue), the LDS read-back and its wait.
    static_assert(!DIRECT || (CHROMA && SX == 2 && SY == 2 && !STRIP420), "DIRECT is built for the two-launch 4:2:0 path");
    constexpr int BY = 64 / BX;                          // block rows per strip
    constexpr int NW = kThreads / 64;                    // waves per workgroup
    constexpr int CW = BX * 8 / SX;                      // chroma samples per strip row
    constexpr int CR = BY * 8 / SY;                      // chroma rows under a strip
    // halo bytes per side.  The tiles that k_chroma_idct's planes are copied into by LDS-DMA carry 16: a tile row is
    // then a whole number of 16-byte chunks and the copy takes 4-5 DMA instructions of 16 B per lane instead of
    // 20-36 of 4 B (an LDS-DMA instruction costs 60-185 cycles to ISSUE whatever it moves: the 20 row transfers of
    // a 32 x 2 strip were 15 % of the strip's wall time).  The tiles the strip fills itself keep 4.
    constexpr int HX = SX == 2 ? ((STRIP420 || (SX == 2 && SY == 1 && BX == 32)) ? 4 : 16) : 0;
    constexpr int HY = SY == 2 ? 1 : 0;
    constexpr int PITCH = (CW + 2 * HX) / 4;             // dwords per LDS row
    constexpr int ROWS = CR + 2 * HY;
    // 4:4:4: no k_chroma_idct launch and no chroma round trip through HBM -- every work-item
    // transforms the Cb and Cr blocks that lie under its luma block itself (same geometry), parks
    // their samples as bytes in LDS ([dword][lane], like k_encode_fused) and then does the luma block
    constexpr bool INTHREAD = CHROMA && SX == 1 && SY == 1;
    constexpr int QROWS = QS * CR + 2;                   // QUAD: sample rows of a stack's tile (halo, QS x CR rows, halo): 34
    constexpr int PLANE = QUAD ? QROWS * PITCH : (CHROMA && !INTHREAD) ? ROWS * PITCH : 1;   // dwords per plane of the tile
    constexpr int SEG_DW = BX * 6;                       // one pixel row of one block row: 24 B per block
    constexpr int CPS = SEG_DW / 4;                      // 16-byte chunks per such segment
    // 4:2:2 (wide strips): the 16 x 2 chroma blocks per plane under a strip are exactly one block per
    // work-item for both planes together; a second, nearly empty pass transforms the 8 neighbour
    // blocks that supply the one-sample halo left and right.  No k_chroma_idct launch, no chroma
    // round trip through HBM (it was 134 of 604 MB at 8192 x 8192).
    constexpr bool IN422 = CHROMA && SX == 2 && SY == 1 && BX == 32;
    // 4:2:0 (wide strips), same idea with a two-dimensional halo: pass 1 transforms the 16 chroma
    // blocks per plane under the strip and the 6 blocks per plane left and right of the three block
    // rows involved (edge column / corner sample: 44 work-items); pass 2 the 16 blocks above and the
    // 16 below per plane, of which only the last / first sample row is wanted -- about a third of a
    // block's arithmetic (idct_block_edge_row).  108 blocks where k_chroma_idct transforms 32 per
    // strip, but no second launch and no chroma samples through HBM.  The neighbours' coefficients are fetched
    // without `nt`: the strips above and below run at the same time on the same XCD (strip s and
    // s + 32 are 8 workgroups apart) and find them in L2.
    // STRIP420 selects it: it wins where the second launch is what costs (one image of up to 4096 x 4096:
    // 27 instead of 31 us there, 12.4 instead of 14.5 at 2048 x 2048), it ties at 8192 x 8192 and loses on
    // batches of narrower images, whose vertical neighbours run on other XCDs and miss in L2 (+ 15 %).
    constexpr bool IN420 = STRIP420 && CHROMA && SX == 2 && SY == 2 && BX == 32;
    constexpr bool INSTRIP = INTHREAD || IN422 || IN420 || QUAD;   // no k_chroma_idct in front of this kernel
    constexpr int NTAB = INSTRIP ? 3 : 1;
    __shared__ __attribute__((aligned(16))) uint32_t coefbuf[NW][DIRECT ? 4 : 64 * 32];  // 8 KiB per wave
    __shared__ __attribute__((aligned(16))) uint32_t stage[NW][BY * SEG_DW]; // one pixel row x BY block rows
    static_assert(!ALIAS || 2 * PLANE <= 64 * 32, "the chroma tile must fit the coefficient buffer");
    __shared__ uint32_t scw[NW][INTHREAD ? 32 * 64 : (ALIAS || QUAD) ? 1 : 2 * PLANE];  // chroma samples under the strip (+ halo) / 4:4:4 stash
    __shared__ uint32_t qtile[QUAD ? QG * 2 * PLANE : 1];   // QUAD: one tile per stack; the window of the wave at position p starts at sample row CR p
    __shared__ float sqw[NW][NTAB][64];                  // modulated table(s): luma (, Cb, Cr)
    // QUAD: per stack, two monotonic counters instead of workgroup barriers.  [0] "ready": a wave has written its samples of
    // this trip into the stack's tile; [1] "done": a wave has read the last sample of this trip that another wave wrote.
    __shared__ uint32_t qsync[QUAD ? 2 * QG : 1];

    const int lane0 = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // strip math stays scalar
    uint32_t *stage_w = stage[wave];
    uint32_t *coef_w = coefbuf[wave];
    // LDS byte address of the wave's coefficient buffer (low 32 bits of the flat shared address)
    const uint32_t coef_lds = __builtin_amdgcn_readfirstlane(
        (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint32_t *)coef_w);
    const int qp = wave % QS, qg = wave / QS;            // QUAD: position in the stack, stack of the workgroup
    uint32_t *qt = qtile + (QUAD ? qg * 2 * PLANE : 0);  // QUAD: this stack's tile
    uint32_t *sc = ALIAS ? coef_w : QUAD ? qt + CR * qp * PITCH : scw[wave];
    const uint32_t sc_lds = __builtin_amdgcn_readfirstlane(
        (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint32_t *)sc);
    float *sq = sqw[wave][0];

    // strip s -> image, strip row (BY block rows), strip column (BX blocks)
    FastDiv fd_tpi, fd_tx, fd_qpi;   // by strips per image, strips per row, stacks per image (QUAD)
    fd_tpi.set((uint32_t)a.tiles_per_image); fd_tx.set((uint32_t)a.tiles_x); fd_qpi.set((uint32_t)max(a.tiles_per_image / QS, 1));
    auto locate = [&](int s, int &img, int &syi, int &sxi) {
        uint32_t rem, col;
        img = (int)fd_tpi.div((uint32_t)s, rem);
        syi = (int)fd_tx.div(rem, col);
        sxi = (int)col;
    };

    // LDS-DMA of the 64 blocks of strip s: instruction i moves 64 x 16 B; slot
    // u = 64 i + lane holds chunk (u & 7) ^ ((b >> 1) & 7) of block b = u >> 3.
    // QUAD: what work-item b of the wave at position qp transforms in the stack's chroma pass (CBW x CBR chroma blocks per plane lie
    // under a strip):
    //   0..31            the strip's own blocks: plane b >> 4, then row-major (b & 15) over CBR rows of CBW columns
    //   32..47           the block row above the stack (first half of the stack's waves) or below it (second half), column
    //                    b & (CBW - 1); the plane is the wave's parity (four waves: 16 columns each) or bit 3 of b (two waves)
    //   48..48+4 CBR-1   left / right neighbours of the own rows: row (b - 48) >> 2, plane bit 1, side bit 0
    //   then 4           the same two columns of the row above (first wave) / below (last wave) the stack: corner samples
    constexpr int CBW = BX / 2, CBR = BY / 2;
    constexpr int QCORN0 = 48 + 4 * CBR, QEND = QCORN0 + 4;
    auto quad_block = [&](int b, int syi, int sxi, int &pl, int &bx, int &by) {
        const int top = syi - qp;                                         // strip row of the stack's first strip
        const int above_row = CBR * top - 1, below_row = CBR * (top + QS);   // chroma block rows (clamped by the caller)
        if (b < 32) { const int idx = b & 15; pl = b >> 4; bx = CBW * sxi + idx % CBW; by = CBR * syi + idx / CBW; }
        else if (b < 48) { pl = BX == 32 ? (qp & 1) : ((b >> 3) & 1); bx = CBW * sxi + (b & (CBW - 1)); by = qp < QS / 2 ? above_row : below_row; }
        else {
            const int j = b < QCORN0 ? b - 48 : b - QCORN0;
            pl = (j >> 1) & 1; bx = (j & 1) ? CBW * sxi + CBW : CBW * sxi - 1;
            by = b < QCORN0 ? CBR * syi + (j >> 2) : (qp == 0 ? above_row : below_row);
        }
    };
    // which: 0 luma; 4:4:4: 1 Cb, 2 Cr; 4:2:2: 1 both chroma planes under the strip, 2 their halo blocks
    auto dma_strip = [&](int s, int lane, int which = 0) {
        int img, syi, sxi;
        locate(s, img, syi, sxi);
        const int16_t *base = a.coef + img * a.coef_stride;
        if constexpr (INTHREAD) {
            if (which) base = a.ccoef[which - 1] + img * a.ccoef_stride[which - 1];
        }
        if constexpr (QUAD) {
            if (which == 3) {   // the wave's chroma pass: block b is what work-item b transforms (quad_block)
                const int uxc = a.pw_c >> 3, uyc = a.ph_c >> 3;
#pragma unroll
                for (int i = 0; i < (QEND + 7) / 8; ++i) {
                    const int b = 8 * i + (lane >> 3);
                    int pl, bx, by;
                    if constexpr (BX == 32) {
                        // the same map as quad_block, decided per instruction where the eight blocks of one are of a kind
                        // (left to the general form every instruction branches per lane: 91 instead of 87 us at 8192 x 8192)
                        const int top = syi - qp, halo_row = qp < 2 ? top - 1 : top + QS;
                        if (i < 4) { pl = i >> 1; bx = 16 * sxi + (b & 15); by = syi; }
                        else if (i < 6) { pl = qp & 1; bx = 16 * sxi + (b & 15); by = halo_row; }
                        else { pl = (b >> 1) & 1; bx = (b & 1) ? 16 * sxi + 16 : 16 * sxi - 1; by = b < 52 ? syi : halo_row; }
                    } else {
                        quad_block(b, syi, sxi, pl, bx, by);
                    }
                    by = min(max(by, 0), uyc - 1);   // missing rows: fetched, not used
                    const int16_t *cbase = a.ccoef[pl] + img * a.ccoef_stride[pl];
                    const uint32_t blk = (bx >= 0 && bx < uxc) ? (uint32_t)by * uxc + bx : 0u;
                    const int c = (lane & 7) ^ ((b >> 1) & 7);
                    lds_dma16_keep(reinterpret_cast<const char *>(cbase) + ((size_t)blk * 128 + 16 * c), coef_lds + 1024 * i);
                }
                return;
            }
        }
        if constexpr (IN420) {
            const int uxc = a.pw_c >> 3, uyc = a.ph_c >> 3;
            const int row_above = max(syi - 1, 0), row_below = min(syi + 1, uyc - 1);   // missing rows: fetched, not used
            if (which == 1) {
                // blocks 0..31: plane b >> 4, column b & 15 of the strip's own chroma row; blocks 32..43: plane
                // (b - 32) / 6, row syi - 1 + ((b - 32) % 6 >> 1), side (b - 32) & 1 (0: column 16 sxi - 1, 1: 16 sxi + 16)
#pragma unroll
                for (int i = 0; i < 6; ++i) {
                    const int b = 8 * i + (lane >> 3);
                    int pl, bx, by;
                    if (i < 4) { pl = i >> 1; bx = 16 * sxi + (b & 15); by = syi; }
                    else {
                        const int idx = min(b - 32, 11), j = idx % 6;
                        pl = idx / 6; bx = (j & 1) ? 16 * sxi + 16 : 16 * sxi - 1;
                        by = min(max(syi - 1 + (j >> 1), 0), uyc - 1);
                    }
                    const int16_t *cbase = a.ccoef[pl] + img * a.ccoef_stride[pl];
                    const uint32_t blk = (bx >= 0 && bx < uxc) ? (uint32_t)by * uxc + bx : 0u;
                    const int c = (lane & 7) ^ ((b >> 1) & 7);
                    lds_dma16_keep(reinterpret_cast<const char *>(cbase) + ((size_t)blk * 128 + 16 * c), coef_lds + 1024 * i);
                }
                return;
            }
            if (which == 2) {   // block b: plane b >> 5, (b >> 4) & 1: 0 the row above, 1 the row below; column b & 15
                if (16 * sxi + 16 <= uxc) {
                    const uint32_t l3 = lane >> 3;
                    const uint32_t ve = l3 * 128 + (((lane & 7) ^ (l3 >> 1)) << 4);
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        const int16_t *cbase = a.ccoef[i >> 2] + img * a.ccoef_stride[i >> 2];
                        const uint32_t blk0 = (uint32_t)(((i >> 1) & 1) ? row_below : row_above) * uxc + 16 * sxi + 8 * (i & 1);
                        const uint64_t sb = reinterpret_cast<uint64_t>(cbase) + ((uint64_t)blk0 << 7);
                        lds_dma16_s_keep(sb, (i & 1) ? ve ^ 64u : ve, coef_lds + 1024 * i);
                    }
                    return;
                }
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int16_t *cbase = a.ccoef[i >> 2] + img * a.ccoef_stride[i >> 2];
                    const int b = 8 * i + (lane >> 3);
                    const int bx = 16 * sxi + (b & 15), by = ((b >> 4) & 1) ? row_below : row_above;
                    const uint32_t blk = bx < uxc ? (uint32_t)by * uxc + bx : 0u;
                    const int c = (lane & 7) ^ ((b >> 1) & 7);
                    lds_dma16_keep(reinterpret_cast<const char *>(cbase) + ((size_t)blk * 128 + 16 * c), coef_lds + 1024 * i);
                }
                return;
            }
        }
        if constexpr (IN422) {
            const int uxc = a.pw_c >> 3, uyc = a.ph_c >> 3;
            if (which == 1) {   // block b of the buffer: plane b >> 5, row (b >> 4) & 1, column b & 15
                if (16 * sxi + 16 <= uxc && 2 * syi + 2 <= uyc) {
                    const uint32_t l3 = lane >> 3;
                    const uint32_t ve = l3 * 128 + (((lane & 7) ^ (l3 >> 1)) << 4);
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        const int16_t *cbase = a.ccoef[i >> 2] + img * a.ccoef_stride[i >> 2];
                        const uint32_t blk0 = (uint32_t)(2 * syi + ((i >> 1) & 1)) * uxc + 16 * sxi + 8 * (i & 1);
                        const uint64_t sb = reinterpret_cast<uint64_t>(cbase) + ((uint64_t)blk0 << 7);
                        lds_dma16_s(sb, (i & 1) ? ve ^ 64u : ve, coef_lds + 1024 * i);
                    }
                    return;
                }
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int16_t *cbase = a.ccoef[i >> 2] + img * a.ccoef_stride[i >> 2];
                    const int b = 8 * i + (lane >> 3);
                    const int bx = 16 * sxi + (b & 15), by = 2 * syi + ((b >> 4) & 1);
                    const uint32_t blk = (bx < uxc && by < uyc) ? (uint32_t)by * uxc + bx : 0u;
                    const int c = (lane & 7) ^ ((b >> 1) & 7);
                    lds_dma16(reinterpret_cast<const char *>(cbase) + ((size_t)blk * 128 + 16 * c), coef_lds + 1024 * i);
                }
                return;
            }
            if (which == 2) {   // block b = 0..7: plane b >> 2, side (b >> 1) & 1 (0 left, 1 right), row b & 1
                const int b = lane >> 3;
                const int16_t *cbase = a.ccoef[b >> 2] + img * a.ccoef_stride[b >> 2];
                const int bx = ((b >> 1) & 1) ? 16 * sxi + 16 : 16 * sxi - 1, by = 2 * syi + (b & 1);
                const uint32_t blk = (bx >= 0 && bx < uxc && by < uyc) ? (uint32_t)by * uxc + bx : 0u;
                const int c = (lane & 7) ^ ((b >> 1) & 7);
                lds_dma16(reinterpret_cast<const char *>(cbase) + ((size_t)blk * 128 + 16 * c), coef_lds);
                return;
            }
        }
        if (sxi * BX + BX <= a.ux && BY * syi + BY <= a.uy) {
            // interior strip (wave-uniform test): the block index is scalar, only the lane's
            // place inside an 8-block group (and its swizzled chunk) is per lane
            const uint32_t l3 = lane >> 3;
            const uint32_t ve = l3 * 128 + (((lane & 7) ^ (l3 >> 1)) << 4);  // even i; odd i: chunk ^ 4
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const uint32_t blk0 = (uint32_t)(BY * syi + i / (BX / 8)) * a.ux + sxi * BX + 8 * (i % (BX / 8));
                const uint64_t sb = reinterpret_cast<uint64_t>(base) + ((uint64_t)blk0 << 7);
                lds_dma16_s(sb, (i & 1) ? ve ^ 64u : ve, coef_lds + 1024 * i);
            }
            return;
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int b = 8 * i + (lane >> 3);  // block within the strip: column b % BX, row b / BX
            const int bx = sxi * BX + (b & (BX - 1)), by = BY * syi + (int)((unsigned)b / BX);
            // blocks outside the plane fetch block 0; the store predicate discards their pixels
            const uint32_t blk = (bx < a.ux && by < a.uy) ? (uint32_t)by * a.ux + bx : 0u;
            const int c = (lane & 7) ^ ((b >> 1) & 7);
            const char *g = reinterpret_cast<const char *>(base) + ((size_t)blk * 128 + 16 * c);
            lds_dma16(g, coef_lds + 1024 * i);
        }
    };

    // The wave's walk: elements k, k + stride, ... of a list of `len` strips.  One list for the launch (strip = first_tile + k),
    // or one list per residue of blockIdx.x mod 8 (xcd_images: the strips of images x, x + 8, ... back to back).
    // QUAD: the WORKGROUP walks stacks of QS strips, QG of them per trip (the host only selects it when every image is a whole
    // number of stacks); the wave at position qp of stack qg takes strip row QS R + qp of stack (R, column c).  All waves make
    // the same number of trips -- the barriers inside the loop are met by everyone; a wave whose stack does not exist
    // (odd number of stacks, last trip) only keeps the others company there.
    const bool by_xcd = !QUAD && a.xcd_images > 0;
    const int xcd = by_xcd ? (int)(blockIdx.x & 7u) : 0;
    const int nwaves = QUAD ? (int)gridDim.x : by_xcd ? (int)(gridDim.x >> 3) * NW : (int)gridDim.x * NW;   // the walk's stride
    FastDiv fd_nw; fd_nw.set((uint32_t)nwaves);
    const int nstacks = QUAD ? a.total_tiles / QS : 0;
    const int len = QUAD ? (nstacks + QG - 1) / QG : by_xcd ? ((a.xcd_images - xcd + 7) >> 3) * a.tiles_per_image : a.total_tiles - a.first_tile;
    auto valid = [&](int k) -> bool { return k < len && (!QUAD || k * QG + qg < nstacks); };
    auto strip_at = [&](int k) -> int {
        if constexpr (QUAD) {
            const int q = min(k * QG + qg, nstacks - 1);
            uint32_t rem, c;
            const int im = (int)fd_qpi.div((uint32_t)q, rem);
            const int R = (int)fd_tx.div(rem, c);
            return im * a.tiles_per_image + (QS * R + qp) * a.tiles_x + (int)c;
        }
        if (!by_xcd) return a.first_tile + k;
        uint32_t rem;
        const int q = (int)fd_tpi.div((uint32_t)k, rem);
        return (xcd + 8 * q) * a.tiles_per_image + (int)rem;
    };
    int k = QUAD ? (int)blockIdx.x : by_xcd ? (int)(blockIdx.x >> 3) * NW + wave : (int)blockIdx.x * NW + wave;
    if constexpr (QUAD) {
        if (threadIdx.x < 2 * QG) qsync[threadIdx.x] = 0;
        __syncthreads();   // the only workgroup barrier of the walk
    }
    if (k >= len) return;
    int s = strip_at(k);
    // DIRECT: the block of the NEXT strip this work-item transforms, requested while the current one is worked on
    uint32_t wn[DIRECT ? 32 : 1];
    auto fetch_block = [&](int st, int ln) {
        int im, sy_, sx_;
        locate(st, im, sy_, sx_);
        const int bx = min(sx_ * BX + (ln & (BX - 1)), a.ux - 1), by = min(BY * sy_ + (int)((unsigned)ln / BX), a.uy - 1);   // blocks outside the plane: pixels never stored
        const uint4 *src = reinterpret_cast<const uint4 *>(a.coef + im * a.coef_stride + ((size_t)by * a.ux + bx) * 64);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const uint4 v = src[i];
            wn[(4 * i + 0) % (DIRECT ? 32 : 1)] = v.x; wn[(4 * i + 1) % (DIRECT ? 32 : 1)] = v.y;
            wn[(4 * i + 2) % (DIRECT ? 32 : 1)] = v.z; wn[(4 * i + 3) % (DIRECT ? 32 : 1)] = v.w;
        }
    };
    if constexpr (DIRECT) fetch_block(s, lane0);
    else if (valid(k)) dma_strip(s, lane0, QUAD ? 3 : INSTRIP ? 1 : 0);
    int img_of_table = -1;
    int stores_behind_dma = 0;  // wave-uniform
#ifdef JA_PHASE_PROFILE
    unsigned long long phase_acc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long t_prev = __builtin_readcyclecounter();
    const unsigned long long t_first = t_prev, r_first = __builtin_amdgcn_s_memrealtime();   // shader cycles / 100 MHz ticks
#endif

    uint32_t trip = 0;   // QUAD: trips this stack has completed (what its counters are compared with)
    for (; k < len; k += nwaves, s = strip_at(min(k, len - 1))) {
        if constexpr (QUAD) {
            if (!valid(k)) continue;   // no stack for this wave's pair in the last trip (the counters are per stack)
        }
        // Launder the lane id once per strip: everything below that depends only on the lane is
        // cheap to recompute, but hoisted out of this loop it would pin ~60 VGPRs for good.
        int lane = lane0;
        asm volatile("" : "+v"(lane));
        const int lbx = lane & (BX - 1), seg = (int)((unsigned)lane / BX);
        int img, syi, sxi;
        locate(s, img, syi, sxi);

        // ---- modulated table (only when the image changes) ----
        if (img != img_of_table) {
            const int qk = lane & 7, qh = lane >> 3;
            sq[lane] = modulate_entry(qk, qh, 0.125f, a.quanta[img * a.quanta_stride + 64 * a.qi + zigzag_of(qk, qh)]);
            if constexpr (INSTRIP) {
#pragma unroll
                for (int pl = 0; pl < 2; ++pl)
                    sqw[wave][1 + pl][lane] = modulate_entry(qk, qh, 0.125f,
                        a.quanta[img * a.quanta_stride + 64 * a.cqi[pl] + zigzag_of(qk, qh)]);
            }
            img_of_table = img;
        }

        // ---- this strip's coefficients: wait for the DMA, read 8 x 16 B (swizzled).  VM
        //      operations retire in issue order and the DMA was issued BEFORE the previous
        //      strip's pixel stores: when that strip took the branch-free store path (exactly
        //      2 store instructions per pixel row) only the DMA has to be waited for, not the
        //      16 stores behind it. ----
        if constexpr (!DIRECT) {
            if (stores_behind_dma == 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
            else if (stores_behind_dma == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        JA_PHASE(0)
#ifndef JA_X_NOPRIO
        // The waves of a SIMD do not advance at the same pace: the scheduler issues the oldest ready wave first, so with
        // equal shares the first wave of a SIMD is done long before the last (8192 x 8192, four strips each: ends between
        // 30 and 72 us, tools/phase_profile.py) and the SIMD spends the end of the launch with one or two waves -- too
        // few to keep it busy.  A wave with more strips left therefore runs at a higher priority: the laggards catch up
        // and all waves of a SIMD leave within a strip of each other (ends between 47 and 66 us).
        {
            uint32_t rr_;
            const int rem = (int)fd_nw.div((uint32_t)(len - 1 - k), rr_);   // strips after this one
            if constexpr (QUAD) __builtin_amdgcn_s_setprio(3);   // see the meeting point below
            else if (rem >= 3) __builtin_amdgcn_s_setprio(3);
            else if (rem == 2) __builtin_amdgcn_s_setprio(2);
            else if (rem == 1) __builtin_amdgcn_s_setprio(1);
            else __builtin_amdgcn_s_setprio(0);
        }
#endif
        uint32_t w[32];
        auto read_block = [&]() {
            const uint4 *cw = reinterpret_cast<const uint4 *>(coef_w) + 8 * lane;
            const int sw = (lane >> 1) & 7;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const uint4 v = cw[i ^ sw];
                w[4 * i + 0] = v.x; w[4 * i + 1] = v.y; w[4 * i + 2] = v.z; w[4 * i + 3] = v.w;
            }
        };
        if constexpr (DIRECT) {
#pragma unroll
            for (int i = 0; i < 32; ++i) w[i] = wn[i % (DIRECT ? 32 : 1)];
        } else {
            read_block();
            // ALIAS: the chroma rows are about to land where the coefficients still are
            if constexpr (ALIAS) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        if constexpr (INTHREAD) {
            // Cb, then Cr: while one plane is transformed the next one's coefficients are on their
            // way into the (single) LDS buffer -- the block has to be in registers before the DMA
            // may overwrite it, hence the lgkmcnt wait
#pragma unroll 1
            for (int pl = 0; pl < 2; ++pl) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                dma_strip(s, lane, pl == 0 ? 2 : 0);
                float g[64];
                idct_block(w, sqw[wave][1 + pl], 128.5f, g);
#pragma unroll
                for (int y = 0; y < 8; ++y)
#pragma unroll
                    for (int d = 0; d < 2; ++d) {
                        uint32_t v = 0;
#pragma unroll
                        for (int i = 0; i < 4; ++i)   // clamp [0, 255] + truncate == saturating convert of floor(v)
                            v = __builtin_amdgcn_cvt_pk_u8_f32(floorf(g[8 * y + 4 * d + i]), i, v);
                        sc[(pl * 16 + 2 * y + d) * 64 + lane] = v;
                    }
                __builtin_amdgcn_sched_barrier(0);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                read_block();
            }
        }

        if constexpr (QUAD) {
            const int uxc = a.pw_c >> 3, uyc = a.ph_c >> 3;
            const int top = syi - qp;
            const bool stack_above = top > 0, stack_below = CBR * (top + QS) < uyc;   // uniform over the stack
            const bool has_left = sxi > 0, has_right = CBW * sxi + CBW < uxc;
            auto pack4 = [](const float *v) -> uint32_t {
                uint32_t d = 0;
#pragma unroll
                for (int i = 0; i < 4; ++i) d = __builtin_amdgcn_cvt_pk_u8_f32(floorf(v[i]), i, d);
                return d;
            };
            auto rep1 = [](float v) -> uint32_t { return __builtin_amdgcn_cvt_pk_u8_f32(floorf(v), 0, 0u) * 0x01010101u; };
            // w holds the chroma pass's block (read at the top); the luma blocks of the strip follow it into the buffer
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            dma_strip(s, lane, 0);
            JA_PHASE(1)
            {
                int pl, bx_, by_;
                quad_block(lane, syi, sxi, pl, bx_, by_);
                float g[64];
#ifdef JA_X_NOCIDCT   // experiment (wrong pixels): the QUAD walk without the arithmetic of its chroma transform
#pragma unroll
                for (int i = 0; i < 64; ++i) g[i] = (float)(w[i & 31] >> (i & 32 ? 16 : 0) & 0xff) + sqw[wave][1 + pl][i];
#else
                idct_block(w, sqw[wave][1 + pl], 128.5f, g);
#endif
                // everyone has read the previous trip's tile before anyone overwrites it -- checked HERE, after the transform:
                // the others signalled "done" at the end of their previous strip, a transform ago, so this rarely waits
#pragma unroll
                for (int i = 0; i < 64; ++i) asm volatile("" : "+v"(g[i]));
                JA_PHASE(2)
                lds_wait_ge(&qsync[2 * qg + 1], QS * trip);
                JA_PHASE(3)
                uint32_t *tile = qt + pl * PLANE;          // row 0: halo above the stack; rows 1 + CR p ...: the wave at position p; row QROWS - 1: halo below
                if (lane < 32) {
                    const int idx = lane & 15;
                    uint32_t *dst = tile + (1 + CR * qp + 8 * (idx / CBW)) * PITCH + 1 + 2 * (idx % CBW);
#pragma unroll
                    for (int y = 0; y < 8; ++y) {
                        dst[y * PITCH] = pack4(&g[8 * y]);
                        dst[y * PITCH + 1] = pack4(&g[8 * y + 4]);
                    }
                } else if (lane < 48) {
                    const bool above = qp < QS / 2;
                    if (above ? stack_above : stack_below) {
                        uint32_t *dst = tile + (above ? 0 : QROWS - 1) * PITCH + 1 + 2 * (lane & (CBW - 1));
                        const float *row = above ? &g[56] : &g[0];   // last row of the block above / first row of the block below
                        dst[0] = pack4(row);
                        dst[1] = pack4(row + 4);
                    }
                } else if (lane < QEND) {
                    const int j = lane < QCORN0 ? lane - 48 : lane - QCORN0, side = j & 1;
                    if (side ? has_right : has_left) {
                        uint32_t *col = tile + (side ? PITCH - 1 : 0);
                        if (lane < QCORN0) {
#pragma unroll
                            for (int y = 0; y < 8; ++y) col[(1 + CR * qp + 8 * (j >> 2) + y) * PITCH] = rep1(side ? g[8 * y] : g[8 * y + 7]);
                        } else if (qp == 0 && stack_above) col[0] = rep1(side ? g[56] : g[63]);
                        else if (qp == QS - 1 && stack_below) col[(QROWS - 1) * PITCH] = rep1(side ? g[0] : g[7]);
                    }
                }
            }
            // the plane's left / right edge: the reference clamps the sample index (decode.swift:4245) -- own rows here,
            // the two halo rows after the barrier (their samples come from other waves)
            const int first_bad = (a.pw_c >> 2) - (sxi * CW - HX) / 4;   // first tile dword past the plane (PITCH - 1 at a full last tile)
            auto fix_columns = [&](uint32_t *row) {
                if (!has_left) row[0] = (row[1] & 0xffu) * 0x01010101u;
                if (first_bad < PITCH) {
                    const uint32_t last = (row[first_bad - 1] >> 24) * 0x01010101u;
                    for (int c = first_bad; c < PITCH; ++c) row[c] = last;
                }
            };
            if (!has_left || first_bad < PITCH) {
                if (lane < 2 * CR) fix_columns(qt + (lane / CR) * PLANE + (1 + CR * qp + lane % CR) * PITCH);
            }
            // this wave's samples are in the tile: arrive, do not wait -- the luma transform and six of the eight pixel rows need
            // only the wave's own chroma rows.  Up to here the wave ran at the top priority (whoever arrives late is waited
            // for by up to three others); from here on at most at priority 2, by strips left like the other walks.
            lds_arrive(&qsync[2 * qg], lane);
            JA_PHASE(4)
            {
                uint32_t rr_;
                const int rem2 = (int)fd_nw.div((uint32_t)(len - 1 - k), rr_);
                if (rem2 >= 2) __builtin_amdgcn_s_setprio(2);
                else if (rem2 == 1) __builtin_amdgcn_s_setprio(1);
                else __builtin_amdgcn_s_setprio(0);
            }
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            read_block();
        }
        // QUAD, after the wait for the stack's tile: the halo rows of the first / last wave.  Image top / bottom: a missing row is
        // the nearest own row (decode.swift:4246); otherwise the row's two edge columns (decode.swift:4245) -- its samples came
        // from other waves.  Only this wave reads the row it repairs.
        auto quad_fix_halo_rows = [&]() {
            if constexpr (QUAD) {
                if (qp == 0 || qp == QS - 1) {
                    const int uyc = a.ph_c >> 3, top = syi - qp;
                    const bool stack_above = top > 0, stack_below = CBR * (top + QS) < uyc, has_left = sxi > 0;
                    const int first_bad = (a.pw_c >> 2) - (sxi * CW - HX) / 4;
                    auto one = [&](int hr, int src, bool missing) {
                        if (missing) {
                            for (int d = lane; d < 2 * PITCH; d += 64) {
                                uint32_t *col = qt + (d >= PITCH ? PLANE + d - PITCH : d);
                                col[hr * PITCH] = col[src * PITCH];
                            }
                        } else if ((!has_left || first_bad < PITCH) && lane < 2) {
                            uint32_t *row = qt + lane * PLANE + hr * PITCH;
                            if (!has_left) row[0] = (row[1] & 0xffu) * 0x01010101u;
                            if (first_bad < PITCH) {
                                const uint32_t last = (row[first_bad - 1] >> 24) * 0x01010101u;
                                for (int c = first_bad; c < PITCH; ++c) row[c] = last;
                            }
                        }
                    };
                    if (qp == 0) one(0, 1, !stack_above);
                    if (qp == QS - 1) one(QROWS - 1, QROWS - 2, !stack_below);
                }
            }
        };
        if constexpr (IN420 && !QUAD) {
            const int uxc = a.pw_c >> 3, uyc = a.ph_c >> 3;
            const bool has_above = syi > 0, has_below = syi + 1 < uyc;        // wave-uniform
            const bool has_left = sxi > 0, has_right = 16 * sxi + 16 < uxc;
            auto pack4 = [](const float *v) -> uint32_t {
                uint32_t d = 0;
#pragma unroll
                for (int i = 0; i < 4; ++i) d = __builtin_amdgcn_cvt_pk_u8_f32(floorf(v[i]), i, d);
                return d;
            };
            auto rep1 = [](float v) -> uint32_t { return __builtin_amdgcn_cvt_pk_u8_f32(floorf(v), 0, 0u) * 0x01010101u; };
            // pass 1: the strip's own chroma blocks (8 sample rows -> tile rows 1..8) and the 12 side blocks
            // (edge column of the own row's neighbours, corner sample of the rows above / below)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            dma_strip(s, lane, 2);
            {
                float g[64];
                const int idx = min(max(lane - 32, 0), 11), j = idx % 6;
                const int pl = lane < 32 ? lane >> 4 : idx / 6;
                idct_block(w, sqw[wave][1 + pl], 128.5f, g);
                if (lane < 32) {
                    uint32_t *dst = sc + pl * PLANE + 1 + 2 * (lane & 15);
#pragma unroll
                    for (int y = 0; y < 8; ++y) {
                        dst[(1 + y) * PITCH] = pack4(&g[8 * y]);
                        dst[(1 + y) * PITCH + 1] = pack4(&g[8 * y + 4]);
                    }
                } else if (lane < 44) {
                    const int rowsel = j >> 1, side = j & 1;
                    const bool ok = (side ? has_right : has_left) && (rowsel == 0 ? has_above : rowsel == 2 ? has_below : true);
                    if (ok) {
                        uint32_t *dst = sc + pl * PLANE + (side ? PITCH - 1 : 0);
                        if (rowsel == 1) {
#pragma unroll
                            for (int y = 0; y < 8; ++y) dst[(1 + y) * PITCH] = rep1(side ? g[8 * y] : g[8 * y + 7]);
                        } else if (rowsel == 0) dst[0] = rep1(side ? g[56] : g[63]);
                        else dst[9 * PITCH] = rep1(side ? g[0] : g[7]);
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            read_block();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            dma_strip(s, lane, 0);
            // pass 2: the blocks above (their last sample row -> tile row 0) and below (first row -> tile row 9)
            {
                const int pl = lane >> 5, below = (lane >> 4) & 1;
                float r[8];
                idct_block_edge_row(w, sqw[wave][1 + pl], 128.5f, !below, r);
                if (below ? has_below : has_above) {
                    uint32_t *dst = sc + pl * PLANE + (below ? 9 * PITCH : 0) + 1 + 2 * (lane & 15);
                    dst[0] = pack4(&r[0]);
                    dst[1] = pack4(&r[4]);
                }
            }
            // image edges: the reference clamps sample indices to the padded plane (decode.swift:4245-4246):
            // first rows (a missing row above / below is the nearest own row), then columns
            if (!has_above || !has_below) {
                for (int d = lane; d < 2 * PITCH; d += 64) {
                    const int pl = d >= PITCH ? 1 : 0, c = d - pl * PITCH;
                    uint32_t *col = sc + pl * PLANE + c;
                    if (!has_above) col[0] = col[PITCH];
                    if (!has_below) col[9 * PITCH] = col[8 * PITCH];
                }
            }
            {
                const int first_bad = (a.pw_c >> 2) - (sxi * CW - HX) / 4;   // first tile dword past the plane
                if (sxi == 0 || first_bad < PITCH) {
                    if (lane < 2 * ROWS) {
                        uint32_t *row = sc + lane * PITCH;
                        if (sxi == 0) row[0] = (row[1] & 0xffu) * 0x01010101u;
                        if (first_bad < PITCH) {
                            const uint32_t last = (row[first_bad - 1] >> 24) * 0x01010101u;
                            for (int c = first_bad; c < PITCH; ++c) row[c] = last;
                        }
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            read_block();
        }
        if constexpr (IN422) {
            // pass 1: the strip's own chroma blocks (lane: plane, block row, block column)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            dma_strip(s, lane, 2);
            {
                const int pl = lane >> 5, r = (lane >> 4) & 1, c = lane & 15;
                float g[64];
                idct_block(w, sqw[wave][1 + pl], 128.5f, g);
                uint32_t *dst = sc + pl * PLANE + 8 * r * PITCH + 1 + 2 * c;
#pragma unroll
                for (int y = 0; y < 8; ++y)
#pragma unroll
                    for (int d = 0; d < 2; ++d) {
                        uint32_t v = 0;
#pragma unroll
                        for (int i = 0; i < 4; ++i) v = __builtin_amdgcn_cvt_pk_u8_f32(floorf(g[8 * y + 4 * d + i]), i, v);
                        dst[y * PITCH + d] = v;
                    }
            }
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            read_block();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            dma_strip(s, lane, 0);
            // pass 2 (8 work-items): the neighbour blocks' edge columns -> the tile's halo dwords
            {
                const int pl = (lane >> 2) & 1, side = (lane >> 1) & 1, r = lane & 1;
                float g[64];
                idct_block(w, sqw[wave][1 + pl], 128.5f, g);
                const bool exists = side ? 16 * sxi + 16 < (a.pw_c >> 3) : sxi > 0;
                if (lane < 8 && exists) {
                    uint32_t *dst = sc + pl * PLANE + 8 * r * PITCH + (side ? PITCH - 1 : 0);
#pragma unroll
                    for (int y = 0; y < 8; ++y) {
                        const uint32_t v = __builtin_amdgcn_cvt_pk_u8_f32(floorf(side ? g[8 * y] : g[8 * y + 7]), 0, 0u);
                        dst[y * PITCH] = v * 0x01010101u;
                    }
                }
            }
            // plane edges: the reference clamps the sample index to the padded plane (decode.swift:4245)
            {
                const int first_bad = (a.pw_c >> 2) - (sxi * CW - HX) / 4;   // first tile dword past the plane
                if (sxi == 0 || first_bad < PITCH) {
                    if (lane < 2 * ROWS) {
                        uint32_t *row = sc + lane * PITCH;
                        if (sxi == 0) row[0] = (row[1] & 0xffu) * 0x01010101u;
                        if (first_bad < PITCH) {
                            const uint32_t last = (row[first_bad - 1] >> 24) * 0x01010101u;
                            for (int c = first_bad; c < PITCH; ++c) row[c] = last;
                        }
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            read_block();
        }

        // ---- chroma samples under the strip (+ halo): one LDS-DMA per row straight into this
        //      wave's LDS tile (lane = dword column); they land during the IDCT.  Row index
        //      clamped by the scalar unit, column index clamped per lane to the padded plane;
        //      the replication a clamped COLUMN needs is patched in LDS on edge strips only. ----
        const int cx0 = sxi * CW, cy0 = syi * CR;
        const int pwd = a.pw_c >> 2;
#ifdef JA_X_NOCTILE   // experiment: no copy of the chroma samples under the strip (what do those small reads cost?)
        if constexpr (false) {
#else
        if constexpr (CHROMA && !INSTRIP) {
#endif
            // rows of a narrow tile are packed RPI to a transfer (the LDS image is lane-linear and the
            // tile rows are contiguous): 12 transfers instead of 36 for a 16 x 4 strip of 4:2:0
            constexpr int RPI = (ROWS % (64 / PITCH) == 0) ? 64 / PITCH : 1;
            constexpr int CHUNKS = PITCH / 4;                 // 16-byte chunks per tile row
            constexpr int SLOTS = 2 * ROWS * CHUNKS;          // both planes
            if (PITCH % 4 == 0 && (a.pw_c & 15) == 0 && a.pw_c >= 16) {
                // 16 bytes per lane: slot u = 64 i + lane is chunk u % CHUNKS of tile row u / CHUNKS (rows of both planes
                // back to back); its LDS address is 16 u -- the tile rows are contiguous.  Chunks are clamped to the plane
                // as a whole (the plane is a whole number of chunks wide here); what a clamped chunk holds is repaired below.
#pragma unroll
                for (int i = 0; i < (SLOTS + 63) / 64; ++i) {
                    const int u = 64 * i + lane;
                    const int r = (int)((unsigned)u / CHUNKS), ch = u - r * CHUNKS;
                    const int pl = r >= ROWS ? 1 : 0;
                    const int gy = min(max(cy0 - HY + r - pl * ROWS, 0), a.ph_c - 1);
                    const int gx = min(max(cx0 - HX + 16 * ch, 0), a.pw_c - 16);
                    const uint8_t *g = (pl ? a.cr : a.cb) + img * a.c_stride + ((size_t)gy * a.pw_c + gx);
                    if (u < SLOTS) lds_dma16_keep(g, sc_lds + 1024 * i);
                }
            } else if constexpr (RPI == 1) {
                const uint32_t coff = 4u * (uint32_t)min(max((cx0 - HX) / 4 + lane, 0), pwd - 1);
                if (lane < PITCH) {
#pragma unroll
                    for (int vr = 0; vr < 2 * ROWS; ++vr) {
                        const int pl = vr >= ROWS ? 1 : 0;
                        const int gy = min(max(cy0 - HY + vr - pl * ROWS, 0), a.ph_c - 1);
                        const uint64_t rowbase = reinterpret_cast<uint64_t>((pl ? a.cr : a.cb) + img * a.c_stride) +
                                                 (uint64_t)((uint32_t)gy * (uint32_t)a.pw_c);
                        lds_dma4_s(rowbase, coff, sc_lds + 4 * PITCH * vr);
                    }
                }
            } else {
                const int rin = (int)((unsigned)lane / PITCH), col = lane - rin * PITCH;
                const uint32_t coff = 4u * (uint32_t)min(max((cx0 - HX) / 4 + col, 0), pwd - 1);
                if (lane < RPI * PITCH) {
#pragma unroll
                    for (int k = 0; k < 2 * ROWS / RPI; ++k) {
                        const int pl = k * RPI >= ROWS ? 1 : 0;   // ROWS % RPI == 0: a transfer never straddles the planes
                        const int gy = min(max(cy0 - HY + k * RPI - pl * ROWS + rin, 0), a.ph_c - 1);
                        const uint64_t planebase = reinterpret_cast<uint64_t>((pl ? a.cr : a.cb) + img * a.c_stride);
                        lds_dma4_s(planebase, (uint32_t)gy * (uint32_t)a.pw_c + coff, sc_lds + 4 * PITCH * RPI * k);
                    }
                }
            }
        }

        // ---- luma: dequantise + IDCT, clamp + truncate (decode.swift:4121-4122), kept as
        //      integer-valued floats for the colour matrix ----
        JA_PHASE(5)
        float yv[64];
#ifdef JA_X_NOIDCT  // experiment: how long is a strip without the IDCT arithmetic?
#pragma unroll
        for (int i = 0; i < 64; ++i) yv[i] = (float)(w[i & 31] >> (i & 32 ? 16 : 0) & 0xff);
#else
        idct_block(w, sq, 128.5f, yv);
#pragma unroll
        for (int i = 0; i < 64; ++i) yv[i] = floorf(__builtin_amdgcn_fmed3f(yv[i], 0.0f, 255.0f));
#endif

        // Pin the IDCT HERE: LLVM otherwise sinks it below the waits / DMA (its results are first
        // used in the colour phase) and the wave would park on the chroma rows before doing any
        // arithmetic instead of letting them land during the IDCT.
#pragma unroll
        for (int i = 0; i < 64; ++i) asm volatile("" : "+v"(yv[i]));
        __builtin_amdgcn_sched_barrier(0);
        JA_PHASE(6)

        // ---- the chroma rows have landed (they are the only VM operations in flight) ----
        if constexpr (CHROMA && !INSTRIP) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const int first_bad = pwd - (cx0 - HX) / 4;          // first tile dword past the plane (>= HX / 4 + 1)
            if ((HX > 0 && sxi == 0) || first_bad < PITCH) {     // wave-uniform: edge strips only
                // the reference clamps the SAMPLE index to the padded plane (decode.swift:4245): whatever the clamped
                // transfers put left of the first / right of the last sample is replaced by that sample
                if (lane < 2 * ROWS) {
                    uint32_t *row = sc + lane * PITCH;
                    if (HX > 0 && sxi == 0) row[HX / 4 - 1] = (row[HX / 4] & 0xffu) * 0x01010101u;
                    if (first_bad < PITCH) {
                        const uint32_t last = (row[first_bad - 1] >> 24) * 0x01010101u;
                        for (int c = first_bad; c < PITCH; ++c) row[c] = last;
                    }
                }
            }
        }
        // ---- the coefficient buffer is consumed: prefetch the next strip into it.  From here to
        //      the end of the strip only stores are issued, so nothing waits on the DMA. ----
        if constexpr (!DIRECT && !ALIAS) {
            if (valid(k + nwaves)) dma_strip(strip_at(k + nwaves), lane, QUAD ? 3 : INSTRIP ? 1 : 0);
        }
        // keep the phases apart (hoisting the chroma LDS reads above the IDCT costs ~70 VGPRs)
        __builtin_amdgcn_sched_barrier(0);
        JA_PHASE(7)

        // ---- chroma rows, produced just in time from the LDS tile ----
        constexpr float inv = 1.0f / (float)((SX == 2 ? 4 : 1) * (SY == 2 ? 4 : 1));
        constexpr float bias = MODE == 1 ? -127.5f : 0.5f;
        // Chroma row j of this block's patch: the LDS reads (hraw) and the conversion + horizontal
        // interpolation, x4 when SX == 2 (hconv), are separate so that the reads can be issued one pixel row
        // ahead of their use -- a wave that waits ~150 cycles for LDS ten times per strip leaves its SIMD to
        // two other waves that are as likely to be waiting themselves.
        auto hraw = [&](int pl, int j, uint32_t (&r)[3]) {
            const uint32_t *row = sc + pl * PLANE + (seg * (8 / SY) + j) * PITCH;
            if constexpr (SX == 2) {
                r[0] = row[HX / 4 - 1 + lbx]; r[1] = row[HX / 4 + lbx]; r[2] = row[HX / 4 + 1 + lbx];
            } else if constexpr (INTHREAD) {   // the block's own samples, parked above
                r[0] = sc[(pl * 16 + 2 * j) * 64 + lane]; r[1] = sc[(pl * 16 + 2 * j + 1) * 64 + lane]; r[2] = 0;
            } else {
                r[0] = row[2 * lbx]; r[1] = row[2 * lbx + 1]; r[2] = 0;
            }
        };
        auto hconv = [&](const uint32_t (&r)[3], float (&o)[8]) {
            if constexpr (SX == 2) {
                const float p[6] = {ubyte<3>(r[0]), ubyte<0>(r[1]), ubyte<1>(r[1]),
                                    ubyte<2>(r[1]), ubyte<3>(r[1]), ubyte<0>(r[2])};
                lerp_row_2x(p, o);
            } else {
                o[0] = ubyte<0>(r[0]); o[1] = ubyte<1>(r[0]); o[2] = ubyte<2>(r[0]); o[3] = ubyte<3>(r[0]);
                o[4] = ubyte<0>(r[1]); o[5] = ubyte<1>(r[1]); o[6] = ubyte<2>(r[1]); o[7] = ubyte<3>(r[1]);
            }
        };
        auto hrow = [&](int pl, int j, float (&o)[8]) {
            uint32_t r[3];
            hraw(pl, j, r);
            hconv(r, o);
        };
        // final chroma value of one pixel from the vertically combined sum v
        auto finish = [&](float v) -> float {
            if constexpr (SX == 1 && SY == 1) return MODE == 1 ? v - 128.0f : v;
            else return floorf(__builtin_fmaf(v, inv, bias));
        };

        float hw[2][3][8];  // SY == 2: patch rows j-1, j, j+1 of both planes (sliding window)
        uint32_t rawn[2][3] = {{0, 0, 0}, {0, 0, 0}};   // LDS dwords of the patch row that is converted next
        if constexpr (QUAD) {
            // pixel rows 1 ... 6 first (own chroma rows only): slot 0 = patch row 1 (kept for pixel row 0), slot 1 = patch row 2
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) { hrow(pl, 1, hw[pl][0]); hrow(pl, 2, hw[pl][1]); hraw(pl, 3, rawn[pl]); }
        } else if constexpr (CHROMA && SY == 2) {
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) { hrow(pl, 0, hw[pl][0]); hrow(pl, 1, hw[pl][1]); hraw(pl, 2, rawn[pl]); }
        } else if constexpr (CHROMA) {
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) hraw(pl, 0, rawn[pl]);
        }

        // ---- store geometry: per pixel row the strip's BY segments are 96 chunks of 16 B; a lane
        //      stores chunk `lane` (and lanes 0..31 also chunk 64 + lane).  Byte offsets relative to
        //      the strip's first pixel are computed once; the row advance is scalar.  Both store
        //      instructions of a row cover whole 128-byte lines (segments are 768 or 384 B). ----
        const int tile_px = min(BX * 8, a.W - BX * 8 * sxi);    // pixels of this strip inside the image
        const int nb = 3 * tile_px;                              // bytes per row segment to write
        const uint32_t pitch = 3u * a.W;
        uint8_t *strip_out = a.out + img * a.out_stride + ((size_t)(8 * BY * syi) * a.W + BX * 8 * sxi) * 3;
        int sg0, sg1;   // segments of chunk `lane` and of chunk 64 + lane (the latter for lanes 0..31)
        if constexpr (BX == 32) { sg0 = lane >= 48 ? 1 : 0; sg1 = 1; }
        else { sg0 = (int)((unsigned)lane / CPS); sg1 = (int)((64u + (unsigned)lane) / CPS); }
        const int j0 = lane - CPS * sg0, j1 = 64 + lane - CPS * sg1;
        const uint32_t voff0 = sg0 * 8u * pitch + 16u * j0, voff1 = sg1 * 8u * pitch + 16u * j1;
        const bool full = 8 * BY * syi + 8 * BY <= a.H && tile_px == BX * 8;   // wave-uniform
        const bool col0 = 16 * j0 < nb, col1 = lane < 32 && 16 * j1 < nb;
        stores_behind_dma = (FAST && full) ? (ALIAS ? 6 : 16) : 0;
        JA_PHASE(8)

        // One pixel row of the strip's BY block rows at a time.  The row's LDS and memory traffic is software-
        // pipelined behind the NEXT row's arithmetic: row y is staged (ds_write) and read back as 16-byte chunks
        // (ds_read) right after its arithmetic, but the chunks are stored only after the arithmetic of row y + 1;
        // the chroma dwords of the next patch row are requested a row (SY == 1) or two (SY == 2) ahead.
        uint4 pv0 = make_uint4(0, 0, 0, 0), pv1 = make_uint4(0, 0, 0, 0);   // chunks of the previous pixel row
        auto put = [&](uint8_t *o, const uint4 &v, int j) {
            if constexpr (FAST) {
                // streaming output, never re-read: non-temporal stores keep it from displacing
                // the chroma planes in L2 / Infinity Cache (-6 % step time)
                store_nt16(o, v);
            } else {
                const uint32_t vv[4] = {v.x, v.y, v.z, v.w};
                for (int k = 0; k < 16; ++k)
                    if (16 * j + k < nb) o[k] = (uint8_t)(vv[k >> 2] >> (8 * (k & 3)));
            }
        };
        auto store_row = [&](int yy) {
            uint8_t *rowp = strip_out + (size_t)yy * pitch;   // scalar
#ifdef JA_X_NOSTORE  // experiment: everything but the global stores
            if (a.W < 0)
#endif
            if (FAST && full) {
                put(rowp + voff0, pv0, j0);
                if (lane < 32) put(rowp + voff1, pv1, j1);
            } else {
                if (col0 && 8 * BY * syi + 8 * sg0 + yy < a.H) put(rowp + voff0, pv0, j0);
                if (col1 && 8 * BY * syi + 8 * sg1 + yy < a.H) put(rowp + voff1, pv1, j1);
            }
        };
        // colour of pixel row y of the work-item's block from its luma samples and the row's chroma values; packed as 24 bytes
        auto colour_row = [&](int y, const float (&cv)[2][8], uint32_t (&d)[6]) {
#pragma unroll
            for (int j = 0; j < 6; ++j) d[j] = 0;
#pragma unroll
            for (int x = 0; x < 8; ++x) {
                const float yy = yv[8 * y + x];
                float c0, c1, c2;
                if constexpr (MODE == 1) {
                    if constexpr (CHROMA) {
                        const float pb = cv[0][x], pr = cv[1][x];
                        // jpeg.swift:441-453, op for op (the 0.0 * c terms are exact no-ops).
                        // v_cvt_pk_u8_f32 rounds to nearest-even and saturates; the reference
                        // clamps and TRUNCATES.  G is floored first.  For R and B the bias
                        // kTruncBias = -0.5 + 2^-10 added to y turns round-to-nearest into
                        // truncation for EVERY (y, c) in [0,255] x [-128,127]: the products
                        // 1.402 c / 1.772 c never come closer than 0.004 to an integer
                        // (verified exhaustively by tests/test_colour_rounding.py).
                        // Nor do they come close enough for the rounding of the product to
                        // matter, so R and B take one FMA each (same test, fused variant).
                        const float yb = yy + kTruncBias;
                        c0 = __builtin_fmaf(1.40200f, pr, yb);
                        // G: floor(fma(m_cr, cr, fma(m_cb, cb, y))) equals the reference's
                        // trunc((y + m_cb cb) + m_cr cr) for every (y, cb, cr) -- all 2^24
                        // triples checked in tests/test_colour_rounding.py (the other fused
                        // association is NOT exact).
                        c1 = floorf(__builtin_fmaf(-0.71414f, pr, __builtin_fmaf(-0.34414f, pb, yy)));
                        c2 = __builtin_fmaf(1.77200f, pb, yb);
                    } else {
                        c0 = c1 = c2 = yy;  // cb = cr = 128: every matrix term is +-0
                    }
                } else {
                    c0 = yy;
                    c1 = CHROMA ? cv[0][x] : 128.0f;
                    c2 = CHROMA ? cv[1][x] : 128.0f;
                }
                // saturating convert of an integer-valued float == clamp [0, 255] + truncate
                d[(3 * x + 0) >> 2] = __builtin_amdgcn_cvt_pk_u8_f32(c0, (3 * x + 0) & 3, d[(3 * x + 0) >> 2]);
                d[(3 * x + 1) >> 2] = __builtin_amdgcn_cvt_pk_u8_f32(c1, (3 * x + 1) & 3, d[(3 * x + 1) >> 2]);
                d[(3 * x + 2) >> 2] = __builtin_amdgcn_cvt_pk_u8_f32(c2, (3 * x + 2) & 3, d[(3 * x + 2) >> 2]);
            }
        };
        // the row's traffic: store the PREVIOUS row's chunks (their LDS read was issued a row ago; prev < 0: there is none),
        // stage this row (LDS ops of one wave execute in order) and read it back as chunks
        auto emit_row = [&](int prev, const uint32_t (&d)[6]) {
            if (prev >= 0) store_row(prev);
            uint2 *sw = reinterpret_cast<uint2 *>(stage_w + seg * SEG_DW + lbx * 6);
            sw[0] = make_uint2(d[0], d[1]);
            sw[1] = make_uint2(d[2], d[3]);
            sw[2] = make_uint2(d[4], d[5]);
            pv0 = *reinterpret_cast<const uint4 *>(stage_w + 4 * lane);
            pv1 = *reinterpret_cast<const uint4 *>(stage_w + 4 * (64 + (lane & 31)));
        };
        if constexpr (QUAD) {
            // Pixel rows in the order 1 2 3 4 5 6 | 0 7: only row 0 of the strip's first block row reads the sample row above
            // the wave's own (patch row 0) and only row 7 of its second block row the one below (patch row 5) -- samples that
            // OTHER waves of the stack produce.  The wave arrived at the "ready" counter before its luma transform and
            // checks it only here, a transform and six pixel rows later.  Slots: hw[.][0] patch row 1 throughout,
            // hw[.][1] / hw[.][2] the sliding pair.
            auto step = [&](int y, int near, int far, int prev) {
                __builtin_amdgcn_sched_barrier(0);
                float cv[2][8];
#pragma unroll
                for (int pl = 0; pl < 2; ++pl)
#pragma unroll
                    for (int x = 0; x < 8; ++x) cv[pl][x] = finish(w31(hw[pl][near][x], hw[pl][far][x]));
                uint32_t d[6];
                colour_row(y, cv, d);
                __builtin_amdgcn_sched_barrier(0);
                emit_row(prev, d);
            };
            step(1, 0, 1, -1);                                    // patch rows 1 (near), 2
            step(2, 1, 0, 1);                                     // 2, 1
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) hconv(rawn[pl], hw[pl][2]);   // patch row 3
            step(3, 1, 2, 2);                                     // 2, 3
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) hraw(pl, 4, rawn[pl]);
            step(4, 2, 1, 3);                                     // 3, 2
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) hconv(rawn[pl], hw[pl][1]);   // patch row 4 (row 2 is dead)
            step(5, 2, 1, 4);                                     // 3, 4
            step(6, 1, 2, 5);                                     // 4, 3
            // ---- the stack's tile is complete: everyone's samples of this trip are in it ----
            JA_PHASE(9)
            lds_wait_ge(&qsync[2 * qg], QS * (trip + 1));
            JA_PHASE(10)
            quad_fix_halo_rows();
            uint32_t raw0[2][3];
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) { hraw(pl, 0, raw0[pl]); hraw(pl, 5, rawn[pl]); }
            // those were this trip's last reads of samples another wave wrote (the LDS performs them before the add)
            lds_arrive(&qsync[2 * qg + 1], lane);
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) hconv(raw0[pl], hw[pl][2]);   // patch row 0 (row 3 is dead)
            step(0, 0, 2, 6);                                     // 1, 0
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) hconv(rawn[pl], hw[pl][2]);   // patch row 5
            step(7, 1, 2, 0);                                     // 4, 5
            __builtin_amdgcn_sched_barrier(0);
            store_row(7);
            ++trip;
        } else {
#pragma unroll
        for (int y = 0; y < 8; ++y) {  // pixel row y of every block row of the strip
            __builtin_amdgcn_sched_barrier(0);
            float cv[2][8];
            if constexpr (CHROMA) {
#pragma unroll
                for (int pl = 0; pl < 2; ++pl) {
                    if constexpr (SY == 2) {
                        // window holds patch rows (y>>1), (y>>1)+1, (y>>1)+2; the nearer row
                        // (middle) weighs 3, the farther one (above for even y, below for odd) 1
                        if ((y & 1) == 1) hconv(rawn[pl], hw[pl][2]);
#pragma unroll
                        for (int x = 0; x < 8; ++x)
                            cv[pl][x] = finish(w31(hw[pl][1][x], hw[pl][(y & 1) ? 2 : 0][x]));
                        if ((y & 1) == 1) {
#pragma unroll
                            for (int x = 0; x < 8; ++x) { hw[pl][0][x] = hw[pl][1][x]; hw[pl][1][x] = hw[pl][2][x]; }
                        }
                    } else {
                        float h[8];
                        hconv(rawn[pl], h);
#pragma unroll
                        for (int x = 0; x < 8; ++x) cv[pl][x] = finish(h[x]);
                    }
                }
            }
            uint32_t d[6];
            colour_row(y, cv, d);
            __builtin_amdgcn_sched_barrier(0);
            // ---- the row's traffic: the previous row's stores, this row's staging, and the request for the chroma
            //      dwords of the next patch row ----
            emit_row(y - 1, d);
            if constexpr (DIRECT) {
                if (y == 4 && valid(k + nwaves)) fetch_block(strip_at(k + nwaves), lane);
            }
            if constexpr (CHROMA) {
                if constexpr (SY == 2) {
                    if ((y & 1) == 1 && y < 7) {
#pragma unroll
                        for (int pl = 0; pl < 2; ++pl) hraw(pl, (y >> 1) + 3, rawn[pl]);
                    }
                    if constexpr (ALIAS) {
                        if (y == 5) {   // the tile has been read for the last time: the next strip's coefficients may land on it
                            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                            if (valid(k + nwaves)) dma_strip(strip_at(k + nwaves), lane, 0);
                        }
                    }
                } else if (y < 7) {
#pragma unroll
                    for (int pl = 0; pl < 2; ++pl) hraw(pl, y + 1, rawn[pl]);
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        store_row(7);
        }
        JA_PHASE(11)
    }
#ifdef JA_PHASE_PROFILE
    // slots 6, 7: the wave's life in shader cycles and in ticks of the constant 100 MHz counter -> effective shader clock
    phase_acc[14] = __builtin_readcyclecounter() - t_first;
    phase_acc[15] = __builtin_amdgcn_s_memrealtime() - r_first;
    if (lane0 == 0 && blockIdx.x * NW + wave < 4096)
    {
        for (int i = 0; i < 16; ++i) g_phase_cycles[(blockIdx.x * NW + wave) * 16 + i] = phase_acc[i];
        unsigned long long *wi = g_wave_info + (blockIdx.x * NW + wave) * 4;
        wi[0] = r_first; wi[1] = r_first + phase_acc[15];
        unsigned hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        wi[2] = hw; wi[3] = xcc & 15u;
    }
#endif
}

inline unsigned blocks_for(size_t n) { return (unsigned)((n + kThreads - 1) / kThreads); }

// Persistent grid = what is resident at once: workgroups per CU (LDS- and VGPR-bound, differs per
// instantiation: 3 for 4:2:0 and grey, 2 for the variants with a full-width chroma tile) x CUs.
template <int SX, int SY, int MODE, bool CHROMA, bool FAST, int BX, bool STRIP420 = false, bool DIRECT = false, bool ALIAS = false, bool QUAD = false>
int resident_workgroups()
{
    static int cached = 0;  // one per instantiation
    if (cached == 0) {
        auto kernel = k_luma_fused<SX, SY, MODE, CHROMA, FAST, BX, STRIP420, DIRECT, ALIAS, QUAD>;
        int per_cu = 0, dev = 0, cus = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, kThreads, 0) != hipSuccess || per_cu < 1) per_cu = 2;
        if (hipGetDevice(&dev) != hipSuccess ||
            hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1) cus = 256;
        cached = per_cu * cus;
    }
    return cached;
}

// development switch: JPEG_AMD_DIRECT=1 selects the four-waves-per-SIMD variant of the 4:2:0 luma kernel
inline bool direct_420()
{
    static const bool v = [] { const char *e = std::getenv("JPEG_AMD_DIRECT"); return e && e[0] == '1'; }();
    return v;
}

// JPEG_AMD_QUAD=0: 4:2:0 images made of whole tile columns and whole stacks take the two launches (or STRIP420) like
// every other image instead of k_luma_fused's QUAD walk (one launch, the four waves of a workgroup sharing a chroma tile)
// JPEG_AMD_QUAD=2 (development switch) also sends images with a partial last tile column / strip row through QUAD, e.g.
// 1920 x 1080: bit-identical, but the half-empty eighth column of strips costs more than the walk gains (512 x 1080p:
// 1 582 against 1 491 us with 16 x 4 strips and two launches, profiles/r02_ab_quad_any_width.txt)
inline int quad_mode()   // 0: off, 1: default rule, 2: see above, 4 (development): the 16 x 4-strip variant where those strips are used
{
    static const int v = [] { const char *e = std::getenv("JPEG_AMD_QUAD"); return e ? std::atoi(e) : 1; }();
    return v;
}

// development switch: JPEG_AMD_XCD_IMAGES=1 gives every image of a batch to one XCD (LumaArgs::xcd_images).  OFF by default:
// it does what it was built for -- k_luma_fused's FETCH_SIZE for 512 x 1080p drops from 3.30 to 2.71 GB, the chroma planes
// are fetched 1.1 instead of 3.3 times (tools/pmc_xcd.sh) -- and the batch takes 1 511 instead of 1 488 us: the repeated
// fetches were served by the Infinity Cache, not by HBM, and cost nothing that matters.
inline bool xcd_images_enabled()
{
    static const bool v = [] { const char *e = std::getenv("JPEG_AMD_XCD_IMAGES"); return e && e[0] == '1'; }();
    return v;
}

// development switch: JPEG_AMD_ALIAS=1 selects the four-waves-per-SIMD 4:2:0 luma kernel whose chroma tile lives inside the
// coefficient buffer.  OFF by default: bit-identical and no faster in sustained runs (tools/ab_band.py --env=JPEG_AMD_ALIAS:
// 102.2 vs 103.1 us at 8192 x 8192, 1 470 vs 1 478 us for 512 x 1080p) -- its waves advance 4/3 slower and the shader
// clock drops from 2.03 to 1.81 GHz (tools/phase_profile.py): the step runs at the package power limit either way.
inline bool alias_420()
{
    static const bool v = [] { const char *e = std::getenv("JPEG_AMD_ALIAS"); return e && e[0] == '1'; }();
    return v;
}

template <int MODE, bool FAST, int BX>
hipError_t launch_luma(hipStream_t stream, int wgs, const LumaArgs &a, int sx, int sy, bool chroma)
{
    // grid = min(work, resident capacity); the per-XCD image partition needs the full, 8-divisible grid
    auto go = [&](auto kernel, int cap) {
        LumaArgs b = a;
#ifdef JA_PHASE_PROFILE   // development aid: JA_GRID_CAP=256 runs one workgroup per CU (a wave alone on its SIMD)
        if (const char *e = std::getenv("JA_GRID_CAP")) cap = std::min(cap, std::atoi(e));
#endif
        const int grid = wgs < cap ? wgs : cap;
        if (grid != cap || (grid & 7) != 0) b.xcd_images = 0;
        hipLaunchKernelGGL(kernel, dim3(grid), dim3(kThreads), 0, stream, b);
    };
#define JA_K(SX_, SY_, CH_) go(k_luma_fused<SX_, SY_, MODE, CH_, FAST, BX>, resident_workgroups<SX_, SY_, MODE, CH_, FAST, BX>());
    if (!chroma) JA_K(1, 1, false)
    else if (sx == 2 && sy == 2 && a.ccoef[0] != nullptr) {   // chroma transformed in the strip walk (IN420 / QUAD)
        if constexpr (FAST) {
            if (a.quad) {
                go(k_luma_fused<2, 2, MODE, true, FAST, BX, true, false, false, true>, resident_workgroups<2, 2, MODE, true, FAST, BX, true, false, false, true>());
                return hipGetLastError();
            }
        }
        if constexpr (BX == 32) go(k_luma_fused<2, 2, MODE, true, FAST, 32, true>, resident_workgroups<2, 2, MODE, true, FAST, 32, true>());
    }
    else if (sx == 2 && sy == 2 && direct_420())
        go(k_luma_fused<2, 2, MODE, true, FAST, BX, false, true>, resident_workgroups<2, 2, MODE, true, FAST, BX, false, true>());
    else if (sx == 2 && sy == 2 && alias_420())
        go(k_luma_fused<2, 2, MODE, true, FAST, BX, false, false, true>, resident_workgroups<2, 2, MODE, true, FAST, BX, false, false, true>());
    else if (sx == 2 && sy == 2) JA_K(2, 2, true)
    else if (sx == 2 && sy == 1) JA_K(2, 1, true)
    else if (sx == 1 && sy == 2) JA_K(1, 2, true)
    else JA_K(1, 1, true)
#undef JA_K
    return hipGetLastError();
}

// Strip shape: 32 x 2 blocks unless 16 x 4 covers the plane with fewer strips (a half-empty strip
// costs as much as a full one: 1920 x 1080 is 7.5 x 68 strips of 32 x 2 but exactly 15 x 34 of
// 16 x 4).  Only the 4:2:0 / 4:4:4 / grey kernels come in both shapes: the 4:2:2 and 4:4:0 chroma
// tiles of a 16 x 4 strip would need 64 row transfers.
inline int strip_width(int ux, int uy, int sx, int sy);
// 4:2:0 with the chroma blocks transformed in the strip walk (IN420) instead of by k_chroma_idct: one
// image, wide strips, at most two rounds of the resident waves' worth of strips (see the kernel)
inline bool strip_chroma_420(const jpeg_amd_layout &L, int n_images)
{
#ifdef JA_X_NO_IN420
    return false;
#endif
    if (L.nplanes != 3 || L.scale_x != 2 || L.scale_y != 2) return false;
    if (strip_width(L.units_x[0], L.units_y[0], 2, 2) != 32) return false;
#ifdef JA_X_FORCE_IN420
    return true;
#endif
    const long strips = (long)((L.units_x[0] + 31) / 32) * ((L.units_y[0] + 1) / 2);
    return n_images == 1 && strips <= 6144;
}
inline int strip_width(int ux, int uy, int sx, int sy)
{
    if (sx != sy) return 32;   // (the 4:2:2 in-thread chroma passes are written for the wide strip)
#ifdef JA_X_FORCE_BX
    return JA_X_FORCE_BX;
#endif
    const long wide = (long)((ux + 31) / 32) * ((uy + 1) / 2), narrow = (long)((ux + 15) / 16) * ((uy + 3) / 4);
    return narrow < wide ? 16 : 32;
}

}  // namespace

#ifdef JA_PHASE_PROFILE
extern "C" int jpeg_amd_debug_phase_cycles(unsigned long long *h_out, size_t n)
{
    return (int)hipMemcpyFromSymbol(h_out, HIP_SYMBOL(g_phase_cycles), n * sizeof(unsigned long long));
}
extern "C" int jpeg_amd_debug_wave_info(unsigned long long *h_out, size_t n)
{
    return (int)hipMemcpyFromSymbol(h_out, HIP_SYMBOL(g_wave_info), n * sizeof(unsigned long long));
}
#endif

bool fused_decode_supported(const jpeg_amd_layout &L, bool cosited)
{
    if (L.precision != 8) return false;
    auto units = [](int size, int stride) { return size / stride + (size % stride != 0 ? 1 : 0); };
    for (int p = 0; p < L.nplanes; ++p) {  // geometry must be the layout-derived one
        if (L.units_x[p] != units(L.width * L.factor_x[p], 8 * L.scale_x)) return false;
        if (L.units_y[p] != units(L.height * L.factor_y[p], 8 * L.scale_y)) return false;
    }
    if (L.nplanes == 1) return true;  // single plane: crop copy whatever the factor (decode.swift:4185)
    if (L.nplanes != 3 || cosited) return false;
    if (L.factor_x[0] != L.scale_x || L.factor_y[0] != L.scale_y) return false;
    if (L.scale_x > 2 || L.scale_y > 2) return false;
    for (int p = 1; p < 3; ++p)
        if (L.factor_x[p] != 1 || L.factor_y[p] != 1) return false;
    return true;
}

size_t fused_decode_scratch_bytes(const jpeg_amd_layout &L, int n_images)
{
    if (L.nplanes == 1 || L.scale_y == 1) return 0;   // grey, 4:4:4, 4:2:2: no intermediate
    if (strip_chroma_420(L, n_images)) return 0;
    const size_t plane = (size_t)64 * L.units_x[1] * L.units_y[1];
    return 2 * ((plane * n_images + 255) & ~(size_t)255);
}

namespace {

// one k_luma_fused launch over strips [la.first_tile, la.total_tiles)
hipError_t launch_luma_any(hipStream_t stream, const LumaArgs &la, int bx, int sx, int sy, bool chroma, bool rgb, bool fast)
{
    // grid == resident capacity (launch_luma clamps).  (Sizing it so that every wave gets the same number of
    // strips -- fewer waves, no thin last round -- measured 4 % slower; 2 instead of 3 workgroups per CU 3.5 %.)
    const int wgs = (la.total_tiles - la.first_tile + 3) / 4;
    if (wgs <= 0) return hipSuccess;
#define JA_L(BX_)                                                                     \
    {                                                                                 \
        if (fast)                                                                     \
            return rgb ? launch_luma<1, true, BX_>(stream, wgs, la, sx, sy, chroma)   \
                       : launch_luma<0, true, BX_>(stream, wgs, la, sx, sy, chroma);  \
        return rgb ? launch_luma<1, false, BX_>(stream, wgs, la, sx, sy, chroma)      \
                   : launch_luma<0, false, BX_>(stream, wgs, la, sx, sy, chroma);     \
    }
    if (bx == 16) JA_L(16)
    JA_L(32)
#undef JA_L
}

// resident waves of k_chroma_idct_persist (4 workgroups per CU by its launch bounds), and a development switch
inline int chroma_persist_waves()
{
    static const int v = [] {
        int per_cu = 0, dev = 0, cus = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_chroma_idct_persist, kThreads, 0) != hipSuccess || per_cu < 1) per_cu = 2;
        if (hipGetDevice(&dev) != hipSuccess ||
            hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1) cus = 256;
        return per_cu * cus * (kThreads / 64);
    }();
    return v;
}
inline bool chroma_persist_enabled()
{
    // OFF by default: measured slower (tools/ab_band.py --env=JPEG_AMD_K1_PERSIST): 103.8 vs 100.4 us per 8192 x 8192 step at
    // three waves per SIMD (168 VGPRs), 125 vs 100 us at four (128 VGPRs, 41 of them spilled) -- like the LDS-DMA
    // variant of round 1 (26-35 vs 25 us).  The one-shot kernel's memory and arithmetic phases do not overlap, but its
    // five waves per SIMD at 94 VGPRs make up for it.
    static const bool v = [] { const char *e = std::getenv("JPEG_AMD_K1_PERSIST"); return e && e[0] == '1'; }();
    return v;
}

// JPEG_AMD_OVERLAP=1 pipelines the two launches of a 4:2:0 / 4:4:0 decode over parts of the call on the context's
// helper streams (OverlapLanes).  OFF by default -- measured on MI355X (tools/ab_band.py --env=JPEG_AMD_OVERLAP):
// the cross-stream event waits cost more than the overlap returns.  One 8192 x 8192 image as two halves 139 us
// against 103 us on one stream, 16 x 2048 x 2048 141 vs 102 us, 512 x 1080p as eight groups 1 506 vs 1 525 us.
inline bool overlap_enabled()
{
    static const bool v = [] { const char *e = std::getenv("JPEG_AMD_OVERLAP"); return e && e[0] == '1'; }();
    return v;
}

}  // namespace

bool fused_decode_wants_lanes() { return overlap_enabled(); }

hipError_t launch_fused_decode(hipStream_t stream, int n_images, const jpeg_amd_layout &L,
                               const PlaneSet &coef, QuantaRef q, bool rgb, void *scratch,
                               uint8_t *d_pixels, size_t pixel_stride, const OverlapLanes *lanes)
{
    const bool chroma = L.nplanes == 3;
    // 4:4:4: k_luma_fused transforms all three planes itself (no intermediate, no first launch)
    // 4:2:2 likewise (the chroma blocks under a strip are one per work-item; see IN422)
    bool inthread = chroma && L.scale_y == 1;
    if (chroma && strip_chroma_420(L, n_images)) inthread = true;   // small single 4:2:0 images too (IN420)
    const bool fast_out = (L.width & 15) == 0 && (pixel_stride & 15) == 0 && (reinterpret_cast<uintptr_t>(d_pixels) & 15) == 0;
    // QUAD: images made of whole 256-pixel tile columns and whole stacks of four 32 x 2-block strips; the last strip row of an
    // image may be partial.  (The kernel also has the walk for 16 x 4 strips -- stacks of two, two stacks per workgroup,
    // 1920 x 1080 is 15 x 17 of them -- but it is 12 % SLOWER than the two launches there: 1 552 against 1 369-1 394 us for
    // 512 x 1080p, JPEG_AMD_QUAD=4.)
    // (With the priorities around the meeting point it also wins where the stacks above and below run on other XCDs:
    // 256 x 512 x 512 93.5 against 100.5 us, 5120 x 5120 44.1 against 49.2, profiles/r02_ab_quad_shapes.txt.  A partial last
    // column does not pay: JPEG_AMD_QUAD=2.)
    bool quad = false;
    int quad_bx = 32;
    if (chroma && L.scale_x == 2 && L.scale_y == 2 && fast_out && quad_mode() != 0) {
        const int sw = strip_width(L.units_x[0], L.units_y[0], 2, 2);
        const bool whole32 = ((L.units_y[0] + 1) / 2) % 4 == 0 && L.units_x[0] % 32 == 0 && (L.width & 255) == 0;
        const bool whole16 = ((L.units_y[0] + 3) / 4) % 2 == 0 && L.units_x[0] % 16 == 0 && (L.width & 127) == 0;
        if (quad_mode() == 2 && ((L.units_y[0] + 1) / 2) % 4 == 0) { quad = true; quad_bx = 32; }
        else if (quad_mode() == 4 && sw == 16 && whole16) { quad = true; quad_bx = 16; }
        else if (whole32) { quad = true; quad_bx = 32; }
    }
    if (quad) inthread = true;
    const bool two_launches = chroma && !inthread;
    LumaArgs la{};
    ChromaArgs ca{};
    const size_t cplane = chroma ? (size_t)64 * L.units_x[1] * L.units_y[1] : 0;
    if (inthread) {
        for (int i = 0; i < 2; ++i) {
            la.ccoef[i] = static_cast<const int16_t *>(coef.ptr[1 + i]);
            la.ccoef_stride[i] = coef.stride[1 + i];
            la.cqi[i] = L.qi[1 + i];
        }
        la.pw_c = 8 * L.units_x[1]; la.ph_c = 8 * L.units_y[1];
    } else if (chroma) {
        const size_t half = (cplane * n_images + 255) & ~(size_t)255;
        for (int i = 0; i < 2; ++i) {
            ca.coef[i] = static_cast<const int16_t *>(coef.ptr[1 + i]);
            ca.coef_stride[i] = coef.stride[1 + i];
            ca.out[i] = static_cast<uint8_t *>(scratch) + i * half;
            ca.qi[i] = L.qi[1 + i];
        }
        ca.out_stride = cplane;
        ca.quanta = q.d_quanta; ca.quanta_stride = q.image_stride;
        ca.ux = L.units_x[1]; ca.first_block = 0; ca.end_block = L.units_x[1] * L.units_y[1];
        la.cb = ca.out[0]; la.cr = ca.out[1]; la.c_stride = cplane;
        la.pw_c = 8 * L.units_x[1]; la.ph_c = 8 * L.units_y[1];
    }
    la.coef = static_cast<const int16_t *>(coef.ptr[0]);
    la.coef_stride = coef.stride[0];
    la.quanta = q.d_quanta; la.quanta_stride = q.image_stride; la.qi = L.qi[0];
    la.ux = L.units_x[0]; la.uy = L.units_y[0];
    la.W = L.width; la.H = L.height;
    la.out = d_pixels; la.out_stride = pixel_stride;
    // unit of work: strip of 32 x 2 (or 16 x 4) luma blocks; persistent waves, 3 per SIMD (VGPR- and LDS-bound)
    const int sx = chroma ? L.scale_x : 1, sy = chroma ? L.scale_y : 1;
    const int bx = quad ? quad_bx : strip_width(la.ux, la.uy, sx, sy), by = 64 / bx;
    la.tiles_x = (la.ux + bx - 1) / bx;
    const int strips_y = (la.uy + by - 1) / by;
    la.tiles_per_image = la.tiles_x * strips_y;
    la.first_tile = 0;
    la.total_tiles = la.tiles_per_image * n_images;
    la.quad = quad ? 1 : 0;
    if (la.total_tiles == 0) return hipSuccess;
    const bool fast = (L.width & 15) == 0 && (pixel_stride & 15) == 0 &&
                      (reinterpret_cast<uintptr_t>(d_pixels) & 15) == 0;

    // one k_chroma_idct launch: blocks [b0, b1) of images [i0, i1)
    auto launch_chroma = [&](hipStream_t st, int i0, int i1, int b0, int b1) -> hipError_t {
#ifdef JA_X_SKIPK1
        return hipSuccess;
#endif
        if (b1 <= b0 || i1 <= i0) return hipSuccess;
        ChromaArgs c = ca;
        for (int i = 0; i < 2; ++i) { c.coef[i] += (size_t)i0 * c.coef_stride[i]; c.out[i] += (size_t)i0 * c.out_stride; }
        c.quanta += (size_t)i0 * c.quanta_stride;
        c.first_block = b0; c.end_block = b1;
        const long units_per_plane = ((long)(b1 - b0) + 63) / 64, total_units = units_per_plane * 2 * (i1 - i0);
        if (chroma_persist_enabled() && total_units >= 2 * chroma_persist_waves() && total_units < 0x7fffffffL) {
            ChromaPersistArgs pa{c, i1 - i0, (int)units_per_plane, (int)total_units};
            hipLaunchKernelGGL(k_chroma_idct_persist, dim3(chroma_persist_waves() / (kThreads / 64)), dim3(kThreads), 0, st, pa);
            return hipGetLastError();
        }
        hipLaunchKernelGGL(k_chroma_idct, dim3(blocks_for(b1 - b0), i1 - i0, 2), dim3(kThreads), 0, st, c);
        return hipGetLastError();
    };
    auto launch_strips = [&](hipStream_t st, int t0, int t1) -> hipError_t {
#ifdef JA_X_SKIPK2
        return hipSuccess;
#endif
        LumaArgs l = la;
        l.first_tile = t0; l.total_tiles = t1;
        return launch_luma_any(st, l, bx, sx, sy, chroma, rgb, fast);
    };

    // ---- parts: groups of images of a batch, or the upper and lower half of one large image ----
    constexpr int kPartTiles = 6144;   // two rounds of the resident waves: below that a part is mostly ramp and tail
    int nparts = 1;
    if (two_launches && lanes && overlap_enabled()) {
        if (n_images > 1) nparts = (int)std::min<long>(std::min<long>(OverlapLanes::kMaxParts, n_images), la.total_tiles / kPartTiles);
        else if (la.total_tiles >= 2 * kPartTiles && strips_y >= 4) nparts = 2;
        if (nparts < 2) nparts = 1;
    }
    if (nparts == 1) {
        if (two_launches) {
            const hipError_t e = launch_chroma(stream, 0, n_images, 0, ca.end_block);
            if (e != hipSuccess) return e;
        }
        // (opt-in) batches of subsampled images: each image's strips on one XCD, see LumaArgs::xcd_images
        if (chroma && n_images >= 32 && xcd_images_enabled()) la.xcd_images = n_images;
        return launch_strips(stream, 0, la.total_tiles);
    }

#define JA_CHECK(expr) do { const hipError_t e_ = (expr); if (e_ != hipSuccess) return e_; } while (0)
    JA_CHECK(hipEventRecord(lanes->entered, stream));
    JA_CHECK(hipStreamWaitEvent(lanes->chroma, lanes->entered, 0));
    JA_CHECK(hipStreamWaitEvent(lanes->luma2, lanes->entered, 0));
    int t_begin[OverlapLanes::kMaxParts + 1];
    if (n_images > 1) {
        for (int p = 0; p <= nparts; ++p) {
            const int img = (int)((long)n_images * p / nparts);
            t_begin[p] = img * la.tiles_per_image;
        }
        for (int p = 0; p < nparts; ++p) {
            JA_CHECK(launch_chroma(lanes->chroma, t_begin[p] / la.tiles_per_image, t_begin[p + 1] / la.tiles_per_image, 0, ca.end_block));
            JA_CHECK(hipEventRecord(lanes->chroma_done[p], lanes->chroma));
        }
    } else {
        // strip rows [0, mid) and [mid, strips_y).  The upper half reads chroma sample rows up to the first row of the
        // chroma block row under strip row `mid` (the bilinear filter reaches one sample down, decode.swift:4243-4257):
        // the first chroma launch goes one block row further than the half; the lower half needs both launches.
        const int mid = strips_y / 2;
        const int c_mid = std::min(L.units_y[1], (mid * by * 8 / sy) / 8 + 1);
        t_begin[0] = 0; t_begin[1] = mid * la.tiles_x; t_begin[2] = la.total_tiles;
        JA_CHECK(launch_chroma(lanes->chroma, 0, 1, 0, c_mid * L.units_x[1]));
        JA_CHECK(hipEventRecord(lanes->chroma_done[0], lanes->chroma));
        JA_CHECK(launch_chroma(lanes->chroma, 0, 1, c_mid * L.units_x[1], ca.end_block));
        JA_CHECK(hipEventRecord(lanes->chroma_done[1], lanes->chroma));
    }
    for (int p = 0; p < nparts; ++p) {
        hipStream_t st = (p & 1) ? lanes->luma2 : stream;
        JA_CHECK(hipStreamWaitEvent(st, lanes->chroma_done[p], 0));
        JA_CHECK(launch_strips(st, t_begin[p], t_begin[p + 1]));
    }
    JA_CHECK(hipEventRecord(lanes->luma2_done, lanes->luma2));
    JA_CHECK(hipStreamWaitEvent(stream, lanes->luma2_done, 0));
#undef JA_CHECK
    return hipSuccess;
}

}  // namespace jpeg_amd
