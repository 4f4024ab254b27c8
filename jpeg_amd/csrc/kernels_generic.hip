// kernels_generic.hip -- Spectral -> Rectangular in ONE launch for the JPEG.Format plug-in path.
//
// Replaces idct() -> interleaved(cosite:) (decode.swift:4154-4165, 4182-4276) for every layout the built-in 8-bit fast
// paths do not take: any precision 1 .. 16 (examples/custom-color/main.swift:41-63 uses 12), one to four planes, centred or
// cosited upsampling, every plane an integer fraction of the image's scale per axis: 1, 1/2, 1/3 or 1/4 (4:1:1, 4:1:0 and the
// factor-3 layouts since round 6; a scale of at most 4).
// Output: the reference's Rectangular, uint16 [H][W][count].  No Planar intermediate in HBM: 128 B per coefficient block in,
// 2 B per sample out ("stops at Rectangular").
//
// EXACT BY CONSTRUCTION: phase A is dct.hpp op for op, as in kernels_stage.hip's k_idct_plane.  Phase B has two forms.  The LITERAL
// one -- per pixel the truncating quotient / remainder of decode.swift:4240-4241, the index clamp to the PADDED plane :4245-4246,
// t = clamp(Float(f) / Float(c)) by a true division :4250-4251 (once per (plane, axis, f), not once per pixel), the two-step
// interpolation :4260-4261 and .rounded() :4264 -- serves the pixels at a plane's edge and the planes whose weights are thirds.
// Everywhere else (round 6) the weights are binary fractions and the samples integers below 2^16: every product and sum of the
// reference's expression is exactly representable, so its value is evaluated in fewer operations on the same exact numbers (see
// the comment in phase B).  None of the ENUMERATED shortcuts of the 8-bit kernels (FMA colour, proofs by exhaustion over 8-bit
// domains) is used here; the quantiser of the encode below is proven for every 16-bit table (tools/verify_div16.hip).
//
// Work decomposition.  The unit is one tile of 128 x TH pixels (TH = 64 when the tile's blocks fit 256 work-items, else 32): a
// workgroup per tile, or -- WALK, for the layouts it pays for (launch_generic_fused) -- resident workgroups that draw their tiles
// from a counter.  Phase A: one 8 x 8 block per work-item (dct.hpp: both passes and transposes are register renames) over the blocks
// of every plane that the tile's pixels read -- for a plane at half resolution that includes a ring of one block, the
// bilinear filter reaches one sample beyond the tile (decode.swift:4243-4257) -- clamp + truncate (decode.swift:4121-4122),
// samples into an LDS tile as uint16.  Phase B: a work-item takes 16 consecutive pixels of a row, gathers every plane's
// sample (copy for planes at the image's scale, decode.swift:4206-4215; bilinear otherwise) and writes 32 * count contiguous
// bytes.  Halo blocks are recomputed by the neighbouring tiles (a 4:2:0 tile transforms 60 chroma blocks per plane for 32 of
// its own): this kernel is the correctness-first generic path, ~2.5 x the instructions per pixel of k_quad420.
#pragma clang fp contract(off)

#include "dct.hpp"
#include "fused_common.hpp"
#include "kernels.hpp"
#include "quantise.hpp"

#include <type_traits>

namespace jpeg_amd {

namespace {

constexpr int kGThreads = 256;
constexpr int GTW = 128;   // tile width in pixels

struct GenPlane {
    const int16_t *coef;
    size_t stride;        // elements between images
    int ux, uy;           // units
    int rx, ry;           // scale / factor per axis: 1 or 2
    int direct;           // factor == scale on both axes, or a single-plane image: cropped copy (decode.swift:4185-4215)
    int qi;
    int ax, bx, cx, ay, by, cy;   // decode.swift:4223-4234
    int lgx, lgy;         // log2 of cx / cy where that is a power of two, else -1 ...
    uint32_t mx, my;      // ... and then ceil(2^32 / c): the quotient by c = 3 or 6 is one v_mul_hi_u32 (exact below 2^26)
    int fastx, fasty;     // the axis is at the image's scale or at exactly half of a scale of 2: the tile's block range and the
                          // shared-sample fetch of phase B are known at compile time; otherwise they follow from (a, b, c)
    float fcy;            // Float(cy), and (cy a power of two) 1 / (s cy) for s = 4, 2, 1: the exact filter's weights and scale come as
    float inv_cy[3];      // kernel arguments -- computed in the kernel they are wave-uniform VALU results, hoisted out of the pass loop
                          // into registers the four-wave build does not have (six spilled, every reload a vmcnt(0))
};
struct GenArgs {
    GenPlane pl[JPEG_AMD_MAX_PLANES];
    const uint16_t *quanta;
    size_t quanta_stride;
    int count, W, H;
    float level, limit;   // decode.swift:4110-4113
    uint16_t *out;
    size_t out_stride;    // elements between images
    int tiles_x, tiles_per_image, total_tiles;   // the call's tiles: image-major, then row-major
    uint32_t *tickets;    // the tiles beyond the first round are DRAWN from this counter: two dwords, zero between calls (the last workgroup
                          // to leave resets them); nullptr: planned
};

// Float.rounded() (to nearest, ties away from zero: decode.swift:4264) of a NON-NEGATIVE float, exactly and without a select:
// with r = trunc(v) the difference d = v - r is exact and lies in [0, 1), 2 d is exact, and trunc(2 d) is 1 exactly when d >= 1/2.
// (roundf() compiles to a sequence around v_cndmask_b32, which issues ten times slower than anything else on gfx950 --
// profiles/r05_probe_valu_classes.txt; the interpolated samples are sums of non-negative products, so v >= 0 always.)
__device__ __forceinline__ float rounded_nonneg(float v)
{
    const float r = __builtin_truncf(v);
    const float d = v - r;
    return r + __builtin_truncf(d + d);
}

// truncating division by c = 1 .. 8 -- Int.quotientAndRemainder, decode.swift:4240.  c a power of two (log2c >= 0): shifts; c = 3 or 6:
// n is negative only for the image's first pixel (n = a = factor - scale > -c), where the quotient truncates to 0, and otherwise
// below 2^26, where mulhi(n, ceil(2^32 / c)) IS the quotient.
__device__ __forceinline__ int div_trunc_small(int n, int c, int log2c, uint32_t magic)
{
    if (log2c >= 0) return (n + ((n >> 31) & (c - 1))) >> log2c;
    return (int)__umulhi((uint32_t)max(n, 0), magic);
}

// Development switches (never defined in the product build): JA_GEN_PHASE (tools/phase_generic.py: wall cycles of every 8th wave per
// phase of its tile), JA_X_GEN_NOLOAD / JA_X_GEN_NOSTORE (the tile without its coefficient loads / its global stores).
#ifdef JA_GEN_PHASE
__device__ unsigned long long g_gen_phase[4096 * 16];
#define GP_DECL unsigned long long gp_acc[16] = {}; unsigned long long gp_prev = __builtin_readcyclecounter(); const unsigned long long gp_first = gp_prev, gp_real = __builtin_amdgcn_s_memrealtime();
#define GP(i) { const unsigned long long n_ = __builtin_readcyclecounter(); gp_acc[i] += n_ - gp_prev; gp_prev = n_; }
#else
#define GP_DECL
#define GP(i)
#endif

// Waves per SIMD = workgroups per CU.  Up to three planes: FOUR (round 6) -- 40 KiB of LDS with the output staged two pixel rows at a
// time, and <= 128 VGPRs since the kernels are built without the SLP vectoriser (jpeg_amd/build.py) and the filter's wave-uniform
// floats arrive as kernel arguments: 4:4:4 16-bit 188 -> 172 us, 4:2:2 12-bit 252 -> 230, 4:1:1 233 -> 222 (8192 x 8192,
// profiles/r06_generic_four_waves.txt).  Four planes: three (42-50 KiB), staged four rows at a time.
template <int COUNT> constexpr int generic_waves_per_simd() { return COUNT == 4 ? 3 : 4; }
template <int TH, int COUNT, bool WALK>
__global__ __launch_bounds__(kGThreads, (generic_waves_per_simd<COUNT>())) void k_generic_fused(GenArgs a_)
{
    constexpr int SROWS = COUNT == 4 ? 4 : 2;   // pixel rows a wave stages at a time
    GP_DECL
    __shared__ __attribute__((aligned(16))) uint16_t tile[kGThreads * 64];   // at most one block per work-item: 32 KiB
    __shared__ float sq[JPEG_AMD_MAX_PLANES][64];                             // modulated tables (natural order)
    __shared__ float tt[JPEG_AMD_MAX_PLANES][2][12];                          // t = clamp(Float(f) / Float(c)), f = -3 .. 8 (at index f + 3)
    __shared__ __attribute__((aligned(16))) uint32_t ostage[kGThreads / 64][SROWS * 64 * COUNT];   // per wave: SROWS rows x 128 px x COUNT samples

    // WALK (round 6): the workgroups of a launch are RESIDENT and walk the call's tiles.  With a workgroup per tile only 2.4 of the
    // four workgroups a CU holds were resident on average (tools/phase_generic.py: 8 192 workgroups x 16.7 us of life in a step of 222 us):
    // the slot of a finished workgroup stays empty for as long as a workgroup lives.  The walk is written so that a trip compiles like
    // the one-tile kernel: the kernel-argument pointer and the work-item id are made opaque at the top of every trip, so nothing that
    // depends on them -- a hundred wave-uniform scalars of four planes -- is hoisted out of the walk into registers the kernel does not
    // have (the first attempts: 303-381 us against 221, tools/exp_patches/r06_generic_decode_tile_walk*.diff).
    typedef const __attribute__((address_space(4))) GenArgs KArgs;
    const int total_tiles = a_.total_tiles;
    // The tiles beyond a workgroup's first are DRAWN from a counter (a.tickets): tiles differ -- those of the image's first and last
    // column, whose edge pixels take the literal filter, cost four times the others, and a plain stride hands a workgroup the SAME
    // tile column on every trip whenever the grid is a multiple of the tiles per row: eight edge tiles in a row made 32 workgroups the
    // last to leave by 90 us (tools/timeline_generic.py) --, XCDs differ in clock, and with tickets whoever is free takes the next
    // tile.  Work-item 0 draws the ticket of the NEXT trip at the top of a trip and publishes it in front of the trip's second
    // barrier.  Without a counter (a call of at most one round) the walk is the plain stride.
    __shared__ int s_next[2];
    // !WALK: a workgroup per tile (grid: tiles x images), one trip, nothing drawn -- the layouts the walk does not pay for (see
    // launch_generic_fused) keep the straight-line kernel: the same body inside a loop of one trip costs them 5-10 %.
    int tile_id = WALK ? (int)blockIdx.x : (int)(blockIdx.y * gridDim.x + blockIdx.x);
    if (tile_id >= total_tiles) return;
#pragma unroll 1
    for (int trip = 0;; ++trip) {
#ifndef JA_X_GEN_NOPRIO
    if constexpr (WALK)
    // A SIMD issues its OLDEST ready wave first: of the four resident workgroups of a CU the youngest would fall behind trip after trip
    // and finish long after the others (without this: mean life of a workgroup 147 us, the step 291).  As in the strip walks, a wave
    // with more tiles left runs at a higher priority, so all of them leave within a tile of each other.
    {
        const int left = (total_tiles - 1 - tile_id) / (int)gridDim.x;   // rounds after this tile's
        if (left >= 3) __builtin_amdgcn_s_setprio(3);
        else if (left == 2) __builtin_amdgcn_s_setprio(2);
        else if (left == 1) __builtin_amdgcn_s_setprio(1);
        else __builtin_amdgcn_s_setprio(0);
    }
#endif
    KArgs *ap = (KArgs *)__builtin_amdgcn_kernarg_segment_ptr();
    if constexpr (WALK) asm volatile("" : "+s"(ap));
    // (WALK: the arguments through the opaque pointer; else the by-value parameter itself)
    const auto *args_ptr = [&]() { if constexpr (WALK) return ap; else return static_cast<const GenArgs *>(&a_); }();
    const auto &a = *args_ptr;
    int t = threadIdx.x;
    if constexpr (WALK) asm volatile("" : "+v"(t));
    uint32_t drawn = 0;
    if constexpr (WALK)
        if (a.tickets != nullptr && t == 0) drawn = __hip_atomic_fetch_add(a.tickets, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int img = WALK ? tile_id / a.tiles_per_image : (int)blockIdx.y, tin = WALK ? tile_id - img * a.tiles_per_image : (int)blockIdx.x;
    // WALK: the tiles of an image are taken in the order "first and last column, then the rest": the column tiles are the expensive
    // ones (their edge pixels take the literal filter: +12 us each), and drawn last they would be the call's tail
    int tyi, txi;
    if (WALK && a.tiles_x >= 3) {
        const int tiles_y = a.tiles_per_image / a.tiles_x, inner = a.tiles_x - 2;
        if (tin < 2 * tiles_y) { tyi = tin >> 1; txi = (tin & 1) ? a.tiles_x - 1 : 0; }
        else { const int r = tin - 2 * tiles_y; tyi = r / inner; txi = 1 + r - tyi * inner; }
    } else { tyi = tin / a.tiles_x; txi = tin - tyi * a.tiles_x; }
    const int x0 = txi * GTW, y0 = tyi * TH;

    // ---- the tile's blocks, plane by plane (wave-uniform scalars) ----
    int bx0[COUNT], by0[COUNT], nbx[COUNT], first[COUNT + 1];
    first[0] = 0;
#pragma unroll
    for (int p = 0; p < COUNT; ++p) {
        const auto &P = a.pl[p];
        int nby;
        if (P.direct || P.fastx) {
            const bool hx = !P.direct && P.rx == 2;
            bx0[p] = hx ? x0 / 16 - 1 : x0 / 8;  nbx[p] = hx ? GTW / 16 + 2 : GTW / 8;
        } else {   // the samples the tile's first and last pixel column read: i of the first, i + 1 of the last (decode.swift:4240-4246)
            const int lo = div_trunc_small(P.ax + P.bx * x0, P.cx, P.lgx, P.mx);
            const int hi = div_trunc_small(P.ax + P.bx * (x0 + GTW - 1), P.cx, P.lgx, P.mx) + 1;
            bx0[p] = lo >> 3; nbx[p] = (hi >> 3) - bx0[p] + 1;
        }
        if (P.direct || P.fasty) {
            const bool hy = !P.direct && P.ry == 2;
            by0[p] = hy ? y0 / 16 - 1 : y0 / 8;  nby = hy ? TH / 16 + 2 : TH / 8;
        } else {
            const int lo = div_trunc_small(P.ay + P.by * y0, P.cy, P.lgy, P.my);
            const int hi = div_trunc_small(P.ay + P.by * (y0 + TH - 1), P.cy, P.lgy, P.my) + 1;
            by0[p] = lo >> 3; nby = (hi >> 3) - by0[p] + 1;
        }
        first[p + 1] = first[p] + nbx[p] * nby;
    }

    // ---- phase A, first half: the work-item's block (one of the blocks any plane contributes to the tile) is REQUESTED before
    //      the tables are built: the two memory latencies at the head of a workgroup -- quanta, then coefficients -- overlap
    //      (round 6; the waves of a tile spent 39 % of their cycles in s_waitcnt / the barriers: profiles/r06_pmc_generic.txt) ----
    bool have = false;
    uint32_t w[32];
    if constexpr (WALK) {
#pragma unroll
        for (int i = 0; i < 32; ++i) w[i] = 0;   // (defined on every path of every trip: an undefined value would be carried around the walk -- and spilled)
    }
    int p = 0, f0 = 0, nx = 1, lbx = 0, lby = 0;
    if (t < first[COUNT]) {
#pragma unroll
        for (int q = 1; q < COUNT; ++q) p += t >= first[q];
        int b0x = bx0[0], b0y = by0[0];
        f0 = first[0]; nx = nbx[0];
#pragma unroll
        for (int q = 1; q < COUNT; ++q)
            if (p == q) { f0 = first[q]; b0x = bx0[q]; b0y = by0[q]; nx = nbx[q]; }
        const int local = t - f0;
        lby = local / nx; lbx = local - lby * nx;
        const int gbx = b0x + lbx, gby = b0y + lby;
        int ux = a.pl[0].ux, uy = a.pl[0].uy;
        const int16_t *cbase = a.pl[0].coef + img * a.pl[0].stride;
#pragma unroll
        for (int q = 1; q < COUNT; ++q)
            if (p == q) { ux = a.pl[q].ux; uy = a.pl[q].uy; cbase = a.pl[q].coef + img * a.pl[q].stride; }
        if (gbx >= 0 && gbx < ux && gby >= 0 && gby < uy) {   // blocks outside the plane are never read (index clamps below)
            have = true;
            const uint4 *src = reinterpret_cast<const uint4 *>(cbase + (size_t)64 * ((size_t)gby * ux + gbx));
#pragma unroll
            for (int i = 0; i < 8; ++i) {
#ifdef JA_X_GEN_NOLOAD   // experiment: the tile without its coefficient loads
                const uint4 v = make_uint4(t + i, gbx, gby, a.W < 0 ? src[i].x : 7u);
#else
                const uint4 v = src[i];
#endif
                w[4 * i + 0] = v.x; w[4 * i + 1] = v.y; w[4 * i + 2] = v.z; w[4 * i + 3] = v.w;
            }
        }
    }
    GP(0)

    // ---- tables: one entry per work-item (decode.swift:3984-4017), and the interpolation weights by a true division ----
    {
        const int p = t >> 6, e = t & 63;
        if (p < COUNT) sq[p][e] = modulate_entry(e & 7, e >> 3, 0.125f, a.quanta[img * a.quanta_stride + 64 * a.pl[p].qi + zigzag_of(e & 7, e >> 3)]);
        if (t < JPEG_AMD_MAX_PLANES * 32) {
            const int pp = t >> 5, axis = (t >> 4) & 1, slot = t & 15, f = slot - 3;
            if (pp < COUNT && slot < 12) {
                const int c = axis ? a.pl[pp].cy : a.pl[pp].cx;
                tt[pp][axis][slot] = fmaxf(0.0f, fminf((float)f / (float)c, 1.0f));   // decode.swift:4250-4251
            }
        }
    }

    GP(1)
    __syncthreads();
    GP(2)
#ifdef JA_GEN_PHASE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    GP(3)
#endif

    // ---- phase A: dequantise + IDCT of the block, samples into the LDS tile ----
    {
        if (have) {
            float g[64];
            idct_block(w, sq[p], a.level, g);
            uint16_t *dst = tile + 64 * f0 + (8 * lby) * (8 * nx) + 8 * lbx;
#pragma unroll
            for (int y = 0; y < 8; ++y) {
                uint32_t s[8];
#pragma unroll
                for (int x = 0; x < 8; ++x) s[x] = clamp_trunc(g[8 * y + x], a.limit);   // decode.swift:4121-4122
                uint4 v;
                v.x = s[0] | (s[1] << 16); v.y = s[2] | (s[3] << 16);
                v.z = s[4] | (s[5] << 16); v.w = s[6] | (s[7] << 16);
                *reinterpret_cast<uint4 *>(dst + y * (8 * nx)) = v;
            }
        }
    }
    GP(4)
    if constexpr (WALK)
        if (t == 0) s_next[(trip + 1) & 1] = a.tickets != nullptr ? (int)gridDim.x + (int)drawn : tile_id + (int)gridDim.x;
    __syncthreads();
    GP(5)

    // ---- phase B: 16 consecutive pixels of a row per work-item and pass ----
#pragma unroll 1
    for (int pass = 0; pass < TH / 32; ++pass) {
        int tl = threadIdx.x;
        asm volatile("" : "+v"(tl));   // opaque per pass: what depends only on the work-item's place would otherwise be hoisted out of the pass loop (~190 VGPRs)
        const int y = y0 + 32 * pass + (tl >> 3);
        const int xb = x0 + 16 * (tl & 7);
        uint32_t o[8 * COUNT];   // 16 pixels x COUNT samples, packed in pairs in the output's order
#pragma unroll
        for (int i = 0; i < 8 * COUNT; ++i) o[i] = 0;
        if (y < a.H && xb < a.W)    // (a work-item without pixels still takes part in the wave's stores below)
#pragma unroll
        for (int p = 0; p < COUNT; ++p) {
            __builtin_amdgcn_sched_barrier(0);   // one plane at a time: left alone, the scheduler issues every plane's LDS reads up front
            const uint16_t *pt = tile + 64 * first[p];
            const int pitch = 8 * nbx[p], ox = 8 * bx0[p], oy = 8 * by0[p];
            uint32_t s[16];
            if (a.pl[p].direct) {   // cropped copy: sample (x, y) of the plane
                const uint4 *row = reinterpret_cast<const uint4 *>(pt + (y - oy) * pitch + (xb - ox));
                const uint4 v0 = row[0], v1 = row[1];
                const uint32_t d[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
#pragma unroll
                for (int i = 0; i < 16; ++i) s[i] = (d[i >> 1] >> (16 * (i & 1))) & 0xffffu;
            } else {
                const auto &P = a.pl[p];
                const int pw = 8 * P.ux, ph = 8 * P.uy;
                const int cols = pitch;
                // the row's vertical position: decode.swift:4240-4251 for y
                const int ny = P.ay + P.by * y;
                const int iy = div_trunc_small(ny, P.cy, P.lgy, P.my), fy = ny - iy * P.cy;
                const int jy = min(iy + 1, ph - 1);
                const float ty = tt[p][1][fy + 3];
                const uint16_t *r0 = pt + (iy - oy) * pitch, *r1 = pt + (jy - oy) * pitch;
                // The horizontal positions of 16 consecutive pixels starting at a multiple of 16 are known at compile time up
                // to the plane's edge clamps: away from the edges the two sample rows are fetched ONCE as whole dwords (the 9 - 17
                // samples the 16 pixels share) and every pixel picks its two by constant index -- the same float operations
                // on the same values as the per-pixel form below (decode.swift:4240-4264), which the threads at the plane's
                // left / right edge keep.  kind: 0 a factor-2 axis, centred (a, b, c = -1, 2, 4: i = (2 x - 1) / 4, f = 3 | 1);
                // 1 a factor-2 axis, cosited (0, 1, 2: i = x / 2, f = x & 1); 2 the axis is at the image's scale (i = x, f = 0).
                // (only for an axis at the image's scale or at half of a scale of 2 -- P.fastx; every other ratio takes the per-pixel form)
                const int kind = P.rx == 1 ? 2 : (P.cx == 4 ? 0 : 1);
                const int seg = tl & 7;
                const bool away = P.fastx && P.lgy >= 0 && (kind == 2 || ((kind == 1 || xb >= 16) && (xb >> 1) + 8 <= pw - 1));
                if (away) {
                    auto pixels = [&](auto K) {
                        constexpr int KIND = decltype(K)::value;
                        constexpr int ND = KIND == 0 ? 6 : KIND == 1 ? 5 : 8;            // dwords per row
                        const int w0 = KIND == 0 ? 8 * seg + 6 : KIND == 1 ? 8 * seg + 8 : 16 * seg;   // first tile column fetched (even)
                        float u0[2 * ND], u1[2 * ND];
#pragma unroll
                        for (int d = 0; d < ND; ++d) {
                            const uint32_t a0 = *reinterpret_cast<const uint32_t *>(r0 + w0 + 2 * d);
                            const uint32_t a1 = *reinterpret_cast<const uint32_t *>(r1 + w0 + 2 * d);
                            u0[2 * d] = (float)(a0 & 0xffffu); u0[2 * d + 1] = (float)(a0 >> 16);
                            u1[2 * d] = (float)(a1 & 0xffffu); u1[2 * d + 1] = (float)(a1 >> 16);
                        }
                        // The weights of these pixels are binary fractions -- horizontally t = 1/4 | 3/4 centred, 0 | 1/2 cosited, 0 at full
                        // scale; vertically f / c with c = 1, 2, 4 or 8 -- and the samples integers below 2^16: every product and sum of
                        // decode.swift:4260-4264 is then an exact binary fraction (at most 16 + 2 + 3 significant bits), i.e. the
                        // reference's value is EXACTLY
                        //     V / (SX * SY),   V = (cx0 u[ia] + cx1 u[ib]) of row i times cy0 + the same of row j times cy1
                        // with small integer weights, whatever the order it is summed in -- and so is this evaluation: one FMA per row
                        // (3 a + b; a + b or 2 a; a), one multiply and one FMA down the column, and .rounded() of the non-negative exact
                        // value as the truncation of V / (SX SY) + 1/2 (exact as well; the conversion truncates).  Six operations per
                        // pixel and plane where the literal sequence takes fourteen; the pixels at a plane's edge keep the literal one.
                        // (This is arithmetic on exactly representable values, not a proof by exhaustion: it holds for every
                        // precision up to 16 bits.  tests/soak_generic.py compares with the literal sequence, evaluated on the CPU.)
                        constexpr float SXW = KIND == 0 ? 4.0f : KIND == 1 ? 2.0f : 1.0f;      // horizontal weights sum to this
                        // vertical: t = clamp(f / c) with c a power of two up to 8 (this path is only taken then): weights f and c - f
                        const float SYW = P.fcy;
                        const float wy1 = (float)max(fy, 0);                                     // (f = a < 0 in the image's first row: t clamps to 0)
                        const float wy0 = SYW - wy1;
                        const float inv = P.inv_cy[KIND];                                        // 1 / (SXW * SYW): a power of two (wave-uniform)
                        (void)SXW;
#pragma unroll
                        for (int i = 0; i < 16; ++i) {
                            // index of sample i_x in the fetched window: centred floor((2 i - 1) / 4) + 2, cosited i / 2, full scale i
                            const int ia = KIND == 0 ? (i == 0 ? 1 : (2 * i - 1) / 4 + 2) : KIND == 1 ? i / 2 : i;
                            const int ib = KIND == 2 ? (i < 15 ? i + 1 : 15) : ia + 1;
                            float h0, h1;
                            if constexpr (KIND == 0) {          // t = 3/4 (even pixels) | 1/4 (odd): 4 x value = u[ia] + 3 u[ib] | 3 u[ia] + u[ib]
                                h0 = (i & 1) ? __builtin_fmaf(u0[ia], 3.0f, u0[ib]) : __builtin_fmaf(u0[ib], 3.0f, u0[ia]);
                                h1 = (i & 1) ? __builtin_fmaf(u1[ia], 3.0f, u1[ib]) : __builtin_fmaf(u1[ib], 3.0f, u1[ia]);
                            } else if constexpr (KIND == 1) {   // t = 0 (even) | 1/2 (odd): 2 x value = 2 u[ia] | u[ia] + u[ib]
                                h0 = (i & 1) ? u0[ia] + u0[ib] : u0[ia] + u0[ia];
                                h1 = (i & 1) ? u1[ia] + u1[ib] : u1[ia] + u1[ia];
                            } else { h0 = u0[ia]; h1 = u1[ia]; (void)ib; }
                            const float V = __builtin_fmaf(h0, wy0, h1 * wy1);
                            s[i] = (uint32_t)__builtin_fmaf(V, inv, 0.5f);
                        }
                    };
                    if (kind == 0) pixels(std::integral_constant<int, 0>{});
                    else if (kind == 1) pixels(std::integral_constant<int, 1>{});
                    else pixels(std::integral_constant<int, 2>{});
                } else
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int x = xb + i;
                    const int nxx = P.ax + P.bx * x;
                    const int ix = div_trunc_small(nxx, P.cx, P.lgx, P.mx), fx = nxx - ix * P.cx;
                    const int jx = min(ix + 1, pw - 1);
                    const float tx = tt[p][0][fx + 3];
                    // tile-local columns; pixels right of the image (never stored) and the weight-0 neighbour of a full-
                    // resolution axis may point past the tile: any finite sample will do there
                    const int lix = min(ix - ox, cols - 1), ljx = min(jx - ox, cols - 1);
                    const float u00 = (float)r0[lix], u01 = (float)r0[ljx];
                    const float u10 = (float)r1[lix], u11 = (float)r1[ljx];
                    const float v0 = u00 * (1.0f - tx) + u01 * tx;   // decode.swift:4260-4261
                    const float v1 = u10 * (1.0f - tx) + u11 * tx;
                    s[i] = (uint32_t)rounded_nonneg(v0 * (1.0f - ty) + v1 * ty);   // :4264
                }
            }
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int e = i * COUNT + p;   // element index within the thread's 16 * COUNT output samples
                o[e >> 1] |= (s[i] & 0xffffu) << (16 * (e & 1));
            }
        }
        // ---- out through a wave-private LDS staging buffer.  A work-item holds 32 COUNT contiguous bytes; storing them itself would
        //      touch 64 different 128-byte lines per instruction, 16 bytes each (partial-line writes: what k_encode_fused once spent a
        //      third of its time on).  Instead the wave's 8 rows x 128 pixels go to LDS four rows at a time and come back
        //      lane-linear: every store instruction writes 64 consecutive 16-byte chunks of whole row segments. ----
        GP(6)
        const int wv = tl >> 6, lane = tl & 63;
        uint32_t *st = ostage[wv];
        const int nvalid = min(GTW, a.W - x0) * COUNT;          // samples of a tile row inside the image
#pragma unroll
        for (int h = 0; h < 8 / SROWS; ++h) {
            if (((tl >> 3) & 7) / SROWS == h) {
                uint4 *mine = reinterpret_cast<uint4 *>(st + ((tl >> 3) & (SROWS - 1)) * (64 * COUNT) + (tl & 7) * (8 * COUNT));
#pragma unroll
                for (int k = 0; k < 2 * COUNT; ++k) mine[k] = make_uint4(o[4 * k], o[4 * k + 1], o[4 * k + 2], o[4 * k + 3]);
            }
            // The lanes read what OTHER lanes of the wave have just written: the wave must reconverge first -- left alone, the
            // compiler duplicates the reads into the branch of the lanes that did not write, and that branch may run before the
            // writers' (seen: the first 32 chunks of every second group of four rows were stale).  The LDS itself executes one
            // wave's operations in order.
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int k = 0; k < (SROWS * COUNT + 3) / 4; ++k) {          // SROWS rows x 16 COUNT chunks, lane-linear
                const int g = 64 * k + lane, row = g / (16 * COUNT), cc = g - row * (16 * COUNT);
                if (SROWS * COUNT % 4 != 0 && g >= SROWS * 16 * COUNT) continue;
                const uint4 v = *reinterpret_cast<const uint4 *>(st + 4 * g);
                const int yy = y0 + 32 * pass + 8 * wv + SROWS * h + row;
#ifdef JA_X_GEN_NOSTORE   // experiment: the tile without its global stores
                if (a.W < 0 || (v.x == 0x12345678u && v.y == 0x9abcdef0u)) {
#else
                if (yy < a.H && 8 * cc < nvalid) {
#endif
                    uint16_t *dst = a.out + img * a.out_stride + ((size_t)yy * a.W + x0) * COUNT + 8 * cc;
                    if (8 * cc + 8 <= nvalid) {
                        *reinterpret_cast<uint4 *>(dst) = v;      // (rows need not be 16-byte aligned: the hardware takes unaligned dwordx4 stores)
                    } else {
                        const uint32_t d[4] = {v.x, v.y, v.z, v.w};
                        for (int e = 0; e < nvalid - 8 * cc; ++e) dst[e] = (uint16_t)(d[e >> 1] >> (16 * (e & 1)));
                    }
                }
            }
            __builtin_amdgcn_wave_barrier();   // ... and the next group's writes stay behind these reads
        }
        GP(7)
    }
    if constexpr (!WALK) break;
    tile_id = s_next[(trip + 1) & 1];
    if (tile_id < total_tiles) continue;   // (wave-uniform)
    if constexpr (WALK) {
        // the last workgroup to leave puts the counter back to zero for the next call (tickets[1] counts the leavers): no memset node
        // in front of the launch (2-3 us and a gap in the stream)
        if (a.tickets != nullptr && t == 0) {
            const uint32_t left_before = __hip_atomic_fetch_add(a.tickets + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (left_before == gridDim.x - 1) {
                __hip_atomic_store(a.tickets, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(a.tickets + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        break;
    }
    }   // the walk.  (No barrier between trips: the next trip's first LDS writes -- the tables -- touch nothing phase B reads but the
        // weights tt, which are the same for every tile of a call; its tile writes come behind its own first barrier.)
#ifdef JA_GEN_PHASE
    {
        const unsigned wid = (blockIdx.y * gridDim.x + blockIdx.x) * 4 + (threadIdx.x >> 6);
        gp_acc[14] = __builtin_readcyclecounter() - gp_first;
        gp_acc[15] = __builtin_amdgcn_s_memrealtime() - gp_real;   // ticks of the constant 100 MHz counter
        {
            unsigned hw_, xcc_;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw_));
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc_));
            gp_acc[12] = gp_real; gp_acc[13] = ((unsigned long long)(xcc_ & 15u) << 32) | hw_;
        }
        if ((threadIdx.x & 63) == 0 && (wid & 7) == 0 && (wid >> 3) < 4096)
            for (int i = 0; i < 16; ++i) g_gen_phase[(wid >> 3) * 16 + i] = gp_acc[i];
    }
#endif
}

// ---------------------------------------------------------------------------------------------------------------------------
// The other direction: Rectangular -> Spectral in ONE launch, replacing decomposed() -> fdct(quanta:) (encode.swift:389-425,
// 199-248) for every layout the 8-bit fused encoder does not take.  The same literal operation sequence as kernels_stage.hip's
// k_decompose + k_fdct_plane: a plane sample is the truncated Float quotient of the sum of the rectangular samples under it
// (indices clamped to the image, encode.swift:403-422); the block is min(limit, .), level-shifted and transformed by dct.hpp's
// fdct_block, every coefficient divided -- a true division -- by its modulated quantum and rounded half away from zero.
//
// One workgroup = a tile of 128 x 32 or 128 x 64 pixels (at the image's scale).  Phase A: the tile's plane samples into LDS, one per
// work-item and step (every rectangular sample is fetched exactly once per plane it belongs to).  Phase B: one 8 x 8 block per
// work-item from the LDS tile -- at most 64 per plane, 256 for four planes at full scale.  Phase C: the blocks' 128 bytes go
// back through the same LDS (chunk-swizzled) and leave as whole runs of neighbouring blocks: every store instruction writes
// 64 consecutive 16-byte chunks instead of 64 chunks 128 bytes apart.
#ifdef JA_GEN_PHASE
}  // namespace
extern "C" int jpeg_amd_debug_gen_phase(unsigned long long *h_out, size_t n)
{
    return (int)hipMemcpyFromSymbol(h_out, HIP_SYMBOL(g_gen_phase), n * sizeof(unsigned long long));
}
namespace {
#endif
constexpr int GEW = 128;   // tile width in pixels; its height GEH is 32, or 64 where the blocks under 64 rows still fit 256 work-items

struct GenEncPlane {
    int16_t *coef;
    size_t stride;        // elements between images
    int ux, uy, rx, ry, qi;
};
struct GenEncArgs {
    GenEncPlane pl[JPEG_AMD_MAX_PLANES];
    const uint16_t *rect;
    size_t rect_stride;   // elements between images
    const uint16_t *quanta;
    size_t quanta_stride;
    int count, W, H;
    float level, limit;   // encode.swift:215-218, :85
    int tiles_x;
};

template <int COUNT, int GEH, int NT = kGThreads>
__global__ __launch_bounds__(NT, (GEH == 64 ? 2 : 4)) void k_generic_encode(GenEncArgs a)
{
    __shared__ __attribute__((aligned(16))) uint16_t raw[GEW * GEH * COUNT];    // the tile of Rectangular; later the blocks on their way out
    extern __shared__ __attribute__((aligned(16))) uint16_t tile[];                // plane tiles: as many samples as the layout's planes have under a tile
    __shared__ float sq[JPEG_AMD_MAX_PLANES][64];                                // modulated tables (natural order, scale 8) ...
    __shared__ float sr[JPEG_AMD_MAX_PLANES][64];                                // ... and their correctly rounded reciprocals
    __shared__ int4 par[JPEG_AMD_MAX_PLANES][2];                                 // per plane: first block, first sample, tile width, units, ratios
    GP_DECL

    const int t = threadIdx.x, img = blockIdx.y;
    const int tyi = blockIdx.x / a.tiles_x, txi = blockIdx.x - tyi * a.tiles_x;
    const int x0 = txi * GEW, y0 = tyi * GEH;
    // ---- phase A1, first half: a tile inside the image REQUESTS its samples before the tables are built -- the two memory latencies at
    //      the head of a workgroup, quanta and samples, overlap (tools/phase_generic.py: they were 42 % of a wave's life in a row) ----
    const uint16_t *rect = a.rect + img * a.rect_stride;
    constexpr int RP = GEW * COUNT;                         // uint16 per tile row
    constexpr int NSTEP = GEH * (RP / 8) / NT;              // 16-byte chunks per work-item
    static_assert(GEH * (RP / 8) % NT == 0, "the tile's chunks divide among the work-items");
    const bool interior = x0 + GEW <= a.W && y0 + GEH <= a.H;   // (wave-uniform)
    uint4 pre[NSTEP];
#pragma unroll
    for (int k = 0; k < NSTEP; ++k) {
        const int i = t + k * NT, row = i / (RP / 8), ch = i - row * (RP / 8);
        pre[k] = make_uint4(0, 0, 0, 0);
        if (interior) pre[k] = *reinterpret_cast<const uint4 *>(rect + ((size_t)a.W * (y0 + row) + x0) * COUNT + 8 * ch);   // (rows need not be 16-byte aligned)
    }
    {
        const int e = t & 63;
        for (int p = t >> 6; p < COUNT; p += NT / 64) {
            const float qv = modulate_entry(e & 7, e >> 3, 8.0f, a.quanta[img * a.quanta_stride + 64 * a.pl[p].qi + zigzag_of(e & 7, e >> 3)]);
            sq[p][e] = qv;
            sr[p][e] = 1.0f / qv;          // IEEE division: RN(1 / q)
        }
    }
    // per plane (wave-uniform): the tile's sample rectangle and where it starts in LDS
    int tw[COUNT], th[COUNT], first[COUNT + 1], fb[COUNT + 1];
    first[0] = 0; fb[0] = 0;
#pragma unroll
    for (int p = 0; p < COUNT; ++p) {
        tw[p] = GEW / a.pl[p].rx; th[p] = GEH / a.pl[p].ry;
        first[p + 1] = first[p] + tw[p] * th[p];
        fb[p + 1] = fb[p] + (tw[p] / 8) * (th[p] / 8);
    }
    if (t < COUNT) {
        par[t][0] = make_int4(fb[t], first[t], tw[t], a.pl[t].ux);
        par[t][1] = make_int4(a.pl[t].uy, a.pl[t].rx, a.pl[t].ry, 0);
    }
    GP(0)
    // ---- phase A1: the tile of Rectangular as it lies in memory -> LDS, 16 bytes per work-item and step; a sample beyond the
    //      image is the nearest one inside (encode.swift:415-417: the box clamps its indices), fetched one at a time ----
#pragma unroll
    for (int k = 0; k < NSTEP; ++k) {
        const int i = t + k * NT, row = i / (RP / 8), ch = i - row * (RP / 8);
        if (interior) *reinterpret_cast<uint4 *>(raw + row * RP + 8 * ch) = pre[k];
    }
    if (!interior)
    for (int i = t; i < GEH * (RP / 8); i += NT) {
        const int row = i / (RP / 8), ch = i - row * (RP / 8);
        const int yy = min(y0 + row, a.H - 1);
        const uint16_t *src = rect + ((size_t)a.W * yy + x0) * COUNT;
        uint4 v;
        if (x0 + (8 * ch + 7) / COUNT <= a.W - 1) {
            v = *reinterpret_cast<const uint4 *>(src + 8 * ch);          // (rows need not be 16-byte aligned)
        } else {
            uint32_t d[4] = {0, 0, 0, 0};
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int el = 8 * ch + e, px = el / COUNT, comp = el - px * COUNT;
                const int xx = min(x0 + px, a.W - 1);
                d[e >> 1] |= (uint32_t)rect[((size_t)a.W * yy + xx) * COUNT + comp] << (16 * (e & 1));
            }
            v = make_uint4(d[0], d[1], d[2], d[3]);
        }
        *reinterpret_cast<uint4 *>(raw + row * RP + 8 * ch) = v;
    }
    GP(1)
    __syncthreads();
    GP(2)
    // ---- phase A2: Rectangular.decomposed() from there into the plane tiles.  The sum of the 1, 2, 4, 8 or 16 samples under a plane
    //      sample (a box of 1, 2 or 4 per axis: encode.swift:403) is below 2^24, so Float(sum) is exact, its quotient by a power of
    //      two is exact, and the truncation of that quotient (encode.swift:404, :422) is the sum shifted right ----
    // (plane by plane: which plane a sample belongs to, and with it the shape of its box, is then the same for the whole wave --
    // per-lane selects are v_cndmask_b32, the one instruction that issues ten times slower than the rest)
    // A work-item takes the samples under EIGHT consecutive pixels of a raw row -- 8, 4 or 2 of them (round 6; one sample per work-item
    // and step, 16-bit LDS reads and writes, was a third of a wave's life: tools/phase_generic.py): the eight pixels are COUNT whole
    // 16-byte chunks (for three components 48 bytes from work-item to work-item: no bank conflict), the component is picked out of
    // the registers by constant index, the results leave as one 16-, 8- or 4-byte write.
#pragma unroll
    for (int p = 0; p < COUNT; ++p) {
        const int rx = a.pl[p].rx, ry = a.pl[p].ry;
        const int lrx = rx == 4 ? 2 : rx == 2 ? 1 : 0, lry = ry == 4 ? 2 : ry == 2 ? 1 : 0;
        const int sh = lrx + lry;
        uint16_t *dst = tile + first[p];
        const int ngroups = 16 * th[p];                       // sixteen groups of eight pixels per plane-tile row
        for (int g = t; g < ngroups; g += NT) {
            const int ly = g >> 4, gx = g & 15;
            const uint16_t *box = raw + (ly * ry) * RP + (8 * gx) * COUNT;
            uint32_t sum[8] = {0, 0, 0, 0, 0, 0, 0, 0};       // per pixel column, down the box
            for (int dy = 0; dy < ry; ++dy) {                 // (wave-uniform trip count)
                uint32_t d[4 * COUNT];
#pragma unroll
                for (int i = 0; i < COUNT; ++i) {
                    const uint4 v = reinterpret_cast<const uint4 *>(box + dy * RP)[i];
                    d[4 * i] = v.x; d[4 * i + 1] = v.y; d[4 * i + 2] = v.z; d[4 * i + 3] = v.w;
                }
#pragma unroll
                for (int px = 0; px < 8; ++px) {
                    const int e = px * COUNT + p;
                    sum[px] += (e & 1) ? d[e >> 1] >> 16 : d[e >> 1] & 0xffffu;
                }
            }
            uint16_t *out = dst + ly * tw[p];
            if (rx == 1) {                                    // (wave-uniform)
                uint4 o;
                o.x = (sum[0] >> sh) | ((sum[1] >> sh) << 16); o.y = (sum[2] >> sh) | ((sum[3] >> sh) << 16);
                o.z = (sum[4] >> sh) | ((sum[5] >> sh) << 16); o.w = (sum[6] >> sh) | ((sum[7] >> sh) << 16);
                *reinterpret_cast<uint4 *>(out + 8 * gx) = o;
            } else if (rx == 2) {
                uint2 o;
                o.x = ((sum[0] + sum[1]) >> sh) | (((sum[2] + sum[3]) >> sh) << 16);
                o.y = ((sum[4] + sum[5]) >> sh) | (((sum[6] + sum[7]) >> sh) << 16);
                *reinterpret_cast<uint2 *>(out + 4 * gx) = o;
            } else {
                *reinterpret_cast<uint32_t *>(out + 2 * gx) =
                    ((sum[0] + sum[1] + sum[2] + sum[3]) >> sh) | (((sum[4] + sum[5] + sum[6] + sum[7]) >> sh) << 16);
            }
        }
    }
    GP(3)
    __syncthreads();
    GP(4)

    // ---- phase B: one block per work-item; a tile with fewer than 256 blocks (96 for 4:2:0) is dealt to ALL waves, a quarter each, so
    //      that the four SIMDs are busy with a partly filled wave each instead of two with full ones ----
    bool have = false;
    int p = 0, gbx = 0, gby = 0;
    // (nearly full waves: as they come; else every wave a RUN of consecutive blocks, the same number each -- neighbouring lanes then
    // read neighbouring 16-byte pieces of a tile row and write blocks 128 bytes apart under the chunk swizzle: block 4 l + w in lane l
    // of wave w, round 5's cut, put every fourth block of a row into a wave and its writes twelve to a bank:
    // SQ_LDS_BANK_CONFLICT 55 M cycles against 16 M of LDS work, profiles/r06_pmc_generic.txt)
    const int per_wave = (fb[COUNT] + NT / 64 - 1) / (NT / 64);
    const int blk = fb[COUNT] > 5 * NT / 8 ? t : ((t & 63) < per_wave ? per_wave * (t >> 6) + (t & 63) : fb[COUNT]);
    if (blk < fb[COUNT]) {
#pragma unroll
        for (int q = 1; q < COUNT; ++q) p += blk >= fb[q];
        // the plane's parameters from a small LDS table indexed by p (a chain of per-lane selects would be 7 (COUNT - 1) v_cndmask_b32)
        const int4 pa = par[p][0], pb = par[p][1];
        const int b0 = pa.x, f0 = pa.y, wpx = pa.z, ux = pa.w, uy = pb.x, rx = pb.y, ry = pb.z;
        const int shift = rx == 4 ? 2 : rx == 2 ? 3 : 4;      // 16 / rx blocks per tile row
        const int local = blk - b0, lby = local >> shift, lbx = local & ((1 << shift) - 1);
        gbx = x0 / (8 * rx) + lbx; gby = y0 / (8 * ry) + lby;
        have = gbx < ux && gby < uy;
        if (have) {
            const uint16_t *src = tile + f0 + (8 * lby) * wpx + 8 * lbx;
            // fdct_block (dct.hpp) with the load in front of its first pass and the quantiser inside its second: a row of samples
            // is fetched when it is transformed, a column of coefficients divided, rounded and packed as soon as it exists
            // (samples and coefficients all at once would hold the kernel at ~200 VGPRs)
            float f[64];  // f[8*k + y]
#pragma unroll
            for (int y = 0; y < 8; ++y) {
                const uint4 v = *reinterpret_cast<const uint4 *>(src + y * wpx);
                const uint32_t d[4] = {v.x, v.y, v.z, v.w};
                float r[8], res[8];
#pragma unroll
                for (int x = 0; x < 8; ++x) r[x] = fminf(a.limit, (float)((d[x >> 1] >> (16 * (x & 1))) & 0xffffu));   // encode.swift:85
                fdct8<true>(r, a.level, res);
#pragma unroll
                for (int k = 0; k < 8; ++k) f[8 * k + y] = res[k];
                __builtin_amdgcn_sched_barrier(0);   // a row at a time (all eight rows' loads up front cost 32 more registers)
            }
            // the coefficients go straight to the staging area (the raw tile is no longer needed: every work-item is past the
            // barrier behind phase A2): int16 at its zigzag position z of block b, 16-byte chunk z / 8 at slot (z / 8) ^ (b & 7)
            uint32_t *mine = reinterpret_cast<uint32_t *>(raw + 64 * blk);
            float zf[64];   // by zigzag index: y1 + copysign(pred(1/2), y1), whose truncation is the rounded coefficient
            // (opaque empty statements pin the program order: left alone, LLVM hoists the table reads of all eight columns over the
            // arithmetic and spills -- kernels_encode.hip's fdct_quantise met the same)
#pragma unroll
            for (int i = 0; i < 64; ++i) asm volatile("" : "+v"(f[i]));
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                __builtin_amdgcn_sched_barrier(0);
                float r[8], res[8];
#pragma unroll
                for (int y = 0; y < 8; ++y) r[y] = f[8 * k + y];
                fdct8<false>(r, 0.0f, res);
#pragma unroll
                for (int h = 0; h < 8; ++h) {
                    // encode.swift:225-240: RN(H / q), rounded half away from zero.  The quotient as y0 = H r; e = fma(-y0, q, H);
                    // y1 = fma(e, r, y0) with r = RN(1 / q) (Markstein's correction step) and the rounding as
                    // trunc(y1 + copysign(pred(1/2), y1)): tools/verify_div16.hip proves by exhaustion on the GPU that the stored
                    // integer equals the reference's for EVERY divisor a 16-bit table can produce (Q = 1 .. 65535, all 64
                    // positions: 1.8 M divisors) and EVERY float numerator below 2^25 -- more than the FDCT of 16-bit samples
                    // reaches (profiles/r06_verify_div16.txt).  Three operations instead of the ten of a true division.
                    const float qq = sq[p][8 * h + k], rr = sr[p][8 * h + k];
                    const float y0 = res[h] * rr;
                    const float e1 = __builtin_fmaf(-y0, qq, res[h]);
                    const float y1 = __builtin_fmaf(e1, rr, y0);
                    zf[zigzag_of(k, h)] = y1 + half_toward(y1);
                }
                // a pair of zigzag neighbours is converted, packed (two conversions, the second into the upper half: quantise.hpp) and
                // written as one dword as soon as the later of its two columns is done; the int16 the reference stores is the low half
                // of the integer either way
                store_ready_pairs(zf, mine, blk & 7, k, std::make_integer_sequence<int, 32>{});
                __builtin_amdgcn_sched_barrier(0);   // one column at a time: eight divisions in flight are register pressure enough
            }
        }
    }
    // ---- phase C: blocks out through LDS (the raw tile is no longer needed); chunk c of block t sits at slot c ^ (t & 7) ----
    const uint4 *stage = reinterpret_cast<const uint4 *>(raw);
    GP(5)
    __syncthreads();
    GP(6)
    // (plane by plane, like phase A2: no per-lane selects)
#pragma unroll
    for (int q = 0; q < COUNT; ++q) {
        const int nbx = tw[q] / 8, rx = a.pl[q].rx, ry = a.pl[q].ry, ux = a.pl[q].ux, uy = a.pl[q].uy;
        const int shift = rx == 4 ? 2 : rx == 2 ? 3 : 4;      // 16 / rx blocks per tile row
        int16_t *coef = a.pl[q].coef + img * a.pl[q].stride;
        const int bx0 = x0 / (8 * rx), by0 = y0 / (8 * ry);
        (void)nbx;
        for (int j = t; j < 8 * (fb[q + 1] - fb[q]); j += NT) {
            const int local = j >> 3, c = j & 7, b = fb[q] + local;
            const int lby = local >> shift, lbx = local & ((1 << shift) - 1);
            const int bx = bx0 + lbx, by = by0 + lby;
            if (bx < ux && by < uy)
                *reinterpret_cast<uint4 *>(coef + (size_t)64 * ((size_t)by * ux + bx) + 8 * c) = stage[8 * b + (c ^ (b & 7))];
        }
    }
    GP(7)
#ifdef JA_GEN_PHASE
    {
        const unsigned wid = (blockIdx.y * gridDim.x + blockIdx.x) * 4 + (threadIdx.x >> 6);
        gp_acc[14] = __builtin_readcyclecounter() - gp_first;
        gp_acc[15] = __builtin_amdgcn_s_memrealtime() - gp_real;
        if ((threadIdx.x & 63) == 0 && (wid & 15) == 0 && (wid >> 4) < 4096)
            for (int i = 0; i < 16; ++i) g_gen_phase[(wid >> 4) * 16 + i] = gp_acc[i];
    }
#endif
}

// blocks of a plane along one axis of a tile of `extent` pixels: at most (k_generic_fused derives the exact range per tile)
int axis_blocks(bool direct, int factor, int scale, int extent)
{
    if (direct || factor == scale) return extent / 8;
    if (scale == 2) return extent / 16 + 2;                       // half of a scale of 2: the tile's blocks + a ring of one
    const int samples = (extent * factor + scale - 1) / scale + 2;   // the samples `extent` pixels read, neighbours included ...
    return (samples + 6) / 8 + 1;                                  // ... starting anywhere inside a block
}
int tile_blocks(const jpeg_amd_layout &L, int th)
{
    int n = 0;
    for (int p = 0; p < L.nplanes; ++p) {
        const bool direct = L.nplanes == 1 || (L.factor_x[p] == L.scale_x && L.factor_y[p] == L.scale_y);
        n += axis_blocks(direct, L.factor_x[p], L.scale_x, GTW) * axis_blocks(direct, L.factor_y[p], L.scale_y, th);
    }
    return n;
}

}  // namespace

bool generic_fused_supported(const jpeg_amd_layout &L)
{
    if (L.nplanes < 1 || L.nplanes > JPEG_AMD_MAX_PLANES || L.precision < 1 || L.precision > 16) return false;
    if (L.scale_x < 1 || L.scale_x > 4 || L.scale_y < 1 || L.scale_y > 4) return false;
    for (int p = 0; p < L.nplanes; ++p) {
        if (L.factor_x[p] < 1 || L.factor_y[p] < 1 || L.factor_x[p] > L.scale_x || L.factor_y[p] > L.scale_y) return false;
        if (L.scale_x % L.factor_x[p] || L.scale_y % L.factor_y[p]) return false;   // (2 in 3, 3 in 4: the staged kernels)
    }
    if (L.width < 1 || L.height < 1 || (long long)L.width * L.height * L.nplanes >= (1LL << 40)) return false;
    return tile_blocks(L, 32) <= kGThreads;
}

hipError_t launch_generic_fused(hipStream_t stream, int n_images, const jpeg_amd_layout &L, const PlaneSet &coef, QuantaRef q,
                                bool cosited, uint32_t *d_walk_counter, uint16_t *d_rect, size_t rect_stride)
{
    GenArgs a{};
    a.count = L.nplanes; a.W = L.width; a.H = L.height;
    a.level = ldexpf(1.0f, L.precision - 1) + 0.5f;   // decode.swift:4110-4113
    a.limit = ldexpf(1.0f, L.precision) - 1.0f;
    a.quanta = q.d_quanta; a.quanta_stride = q.image_stride;
    a.out = d_rect; a.out_stride = rect_stride;
    for (int p = 0; p < L.nplanes; ++p) {
        GenPlane &P = a.pl[p];
        P.coef = static_cast<const int16_t *>(coef.ptr[p]); P.stride = coef.stride[p];
        P.ux = L.units_x[p]; P.uy = L.units_y[p]; P.qi = L.qi[p];
        P.rx = L.scale_x / L.factor_x[p]; P.ry = L.scale_y / L.factor_y[p];
        P.direct = (L.nplanes == 1) || (L.factor_x[p] == L.scale_x && L.factor_y[p] == L.scale_y);
        if (cosited) {  // decode.swift:4223-4234
            P.ax = 0; P.ay = 0; P.bx = L.factor_x[p]; P.by = L.factor_y[p]; P.cx = L.scale_x; P.cy = L.scale_y;
        } else {
            P.ax = L.factor_x[p] - L.scale_x; P.ay = L.factor_y[p] - L.scale_y;
            P.bx = 2 * L.factor_x[p]; P.by = 2 * L.factor_y[p]; P.cx = 2 * L.scale_x; P.cy = 2 * L.scale_y;
        }
        auto log2_of = [](int c) { return c == 1 ? 0 : c == 2 ? 1 : c == 4 ? 2 : c == 8 ? 3 : -1; };
        P.lgx = log2_of(P.cx); P.lgy = log2_of(P.cy);
        P.mx = (uint32_t)((0x100000000ull + (uint64_t)P.cx - 1) / (uint64_t)P.cx); P.my = (uint32_t)((0x100000000ull + (uint64_t)P.cy - 1) / (uint64_t)P.cy);
        P.fastx = P.rx == 1 || (P.rx == 2 && L.scale_x == 2);
        P.fasty = P.ry == 1 || (P.ry == 2 && L.scale_y == 2);
        P.fcy = (float)P.cy;
        for (int k = 0; k < 3; ++k) P.inv_cy[k] = P.lgy >= 0 ? ldexpf(k == 0 ? 0.25f : k == 1 ? 0.5f : 1.0f, -P.lgy) : 0.0f;
    }
    if (n_images == 0) return hipSuccess;
    const int th = tile_blocks(L, 64) <= kGThreads ? 64 : 32;
    a.tiles_x = (L.width + GTW - 1) / GTW;
    a.tiles_per_image = a.tiles_x * ((L.height + th - 1) / th);
    if ((long long)a.tiles_per_image * n_images > 0x7fffffffLL) return hipErrorInvalidValue;
    a.total_tiles = a.tiles_per_image * n_images;
    // The walk (resident workgroups, tickets) or a workgroup per tile?  Measured on nine layouts at 8192 x 8192 (profiles/r06_generic_walk.txt):
    // the walk wins where a tile is long and held up by its own latencies -- 64-row tiles whose subsampled planes all take the exact
    // filter: 4:2:0 centred or cosited, 213 -> 190 us at 12 bits, 197 -> 177 at 8 -- and loses where the kernel is close to a
    // throughput limit and the dispatcher's natural stagger of the workgroups is worth more than their residency (4:4:4 16-bit at
    // 4.7 TB/s: 172 -> 228 us; the layouts on the literal filter, 4:1:1, 4:1:0, thirds: 220 -> 240).  Those keep a workgroup per tile:
    // the same kernel with a grid of all tiles makes one trip and draws nothing.
    bool subsampled = false, all_exact = true;
    for (int p = 0; p < L.nplanes; ++p)
        if (!a.pl[p].direct) { subsampled = true; all_exact = all_exact && a.pl[p].fastx && a.pl[p].fasty && a.pl[p].lgy >= 0; }
    const bool walk = th == 64 && subsampled && all_exact;
#define JA_GK(K_)                                                                                                       \
    {                                                                                                                   \
        const int cap = resident_workgroups_of<K_>(3);                                                                  \
        a.tickets = a.total_tiles > cap ? d_walk_counter : nullptr;   /* (two dwords, zero between calls: the kernel resets them) */ \
        hipLaunchKernelGGL(K_, dim3(a.total_tiles < cap ? a.total_tiles : cap), dim3(kGThreads), 0, stream, a);          \
    }
#define JA_G(TH_, C_)                                                                                                   \
    {                                                                                                                   \
        /* (a call of at most one round keeps the straight-line kernel: nothing to walk) */                             \
        if (TH_ == 64 && walk && a.total_tiles > resident_workgroups_of<k_generic_fused<64, C_, true>>(3))              \
            JA_GK((k_generic_fused<64, C_, true>))                                                                      \
        else hipLaunchKernelGGL((k_generic_fused<TH_, C_, false>), dim3(a.tiles_per_image, n_images), dim3(kGThreads), 0, stream, a); \
    }
#define JA_GC(TH_)                               \
    switch (L.nplanes) {                         \
    case 1: JA_G(TH_, 1); break;                 \
    case 2: JA_G(TH_, 2); break;                 \
    case 3: JA_G(TH_, 3); break;                 \
    default: JA_G(TH_, 4); break;                \
    }
    if (th == 64) { JA_GC(64) } else { JA_GC(32) }
#undef JA_GC
#undef JA_G
#undef JA_GK
    return hipGetLastError();
}

bool generic_encode_supported(const jpeg_amd_layout &L)
{
    // the box of decomposed() spans 1, 2 or 4 samples per axis here (a box of 3 would need tiles 24 k pixels wide and a true
    // division of its sum: those layouts take the staged kernels)
    for (int p = 0; p < L.nplanes; ++p)
        if (L.factor_x[p] >= 1 && L.factor_y[p] >= 1 && (L.scale_x / L.factor_x[p] == 3 || L.scale_y / L.factor_y[p] == 3)) return false;
    return generic_fused_supported(L) && (long long)L.width * L.height * L.nplanes < (1LL << 40);
}

hipError_t launch_generic_encode(hipStream_t stream, int n_images, const jpeg_amd_layout &L, const uint16_t *d_rect, size_t rect_stride,
                                 QuantaRef q, const PlaneSetMut &coef)
{
    GenEncArgs a{};
    a.count = L.nplanes; a.W = L.width; a.H = L.height;
    a.level = ldexpf(1.0f, L.precision - 1) * 8.0f;   // encode.swift:215-218: no +0.5 in the forward level shift
    a.limit = ldexpf(1.0f, L.precision) - 1.0f;
    a.rect = d_rect; a.rect_stride = rect_stride;
    a.quanta = q.d_quanta; a.quanta_stride = q.image_stride;
    int need_x = 0, need_y = 0;                       // the padded planes, in pixels at the image's scale
    for (int p = 0; p < L.nplanes; ++p) {
        GenEncPlane &P = a.pl[p];
        P.coef = static_cast<int16_t *>(coef.ptr[p]); P.stride = coef.stride[p];
        P.ux = L.units_x[p]; P.uy = L.units_y[p]; P.qi = L.qi[p];
        P.rx = L.scale_x / L.factor_x[p]; P.ry = L.scale_y / L.factor_y[p];   // the response of the box, encode.swift:403
        need_x = max(need_x, 8 * P.ux * P.rx); need_y = max(need_y, 8 * P.uy * P.ry);
    }
    if (n_images == 0 || need_x == 0 || need_y == 0) return hipSuccess;
    // tile height: 32 rows.  64 rows would fill the workgroup better (4:2:0: 192 blocks instead of 96 under 32 rows)
    int blocks64 = 0, blocks32 = 0;
    for (int p = 0; p < L.nplanes; ++p) {
        blocks64 += (GEW / (8 * a.pl[p].rx)) * (64 / (8 * a.pl[p].ry));
        blocks32 += (GEW / (8 * a.pl[p].rx)) * (32 / (8 * a.pl[p].ry));
    }
    (void)blocks32;
    // (measured: 64-row tiles are SLOWER -- 8192 x 8192 12-bit 4:2:0 472 against 403 us, 4:2:2 923 against 490 -- although they fill
    // the workgroup's lanes: fewer, larger workgroups hide less of each other's barriers.  JA_X_GENERIC_ENCODE_64 builds them.)
#ifdef JA_X_GENERIC_ENCODE_64
    const int geh = (L.nplanes <= 3 && blocks64 <= kGThreads) ? 64 : 32;
#else
    const int geh = 32;
    (void)blocks64;
#endif
    a.tiles_x = (need_x + GEW - 1) / GEW;
    const dim3 grid(a.tiles_x * ((need_y + geh - 1) / geh), n_images);
    size_t tile_bytes = 0;
    for (int p = 0; p < L.nplanes; ++p) tile_bytes += (size_t)2 * (GEW / a.pl[p].rx) * (geh / a.pl[p].ry);
#ifdef JA_X_GENERIC_ENCODE_64
#define JA_GE(C_) do { if (geh == 64) hipLaunchKernelGGL((k_generic_encode<C_, 64>), grid, dim3(kGThreads), tile_bytes, stream, a); \
                       else hipLaunchKernelGGL((k_generic_encode<C_, 32>), grid, dim3(kGThreads), tile_bytes, stream, a); } while (0)
#else
#ifdef JA_X_GENC_NT
#define JA_GE(C_) do { if (blocks32 <= JA_X_GENC_NT) hipLaunchKernelGGL((k_generic_encode<C_, 32, JA_X_GENC_NT>), grid, dim3(JA_X_GENC_NT), tile_bytes, stream, a); \
                       else hipLaunchKernelGGL((k_generic_encode<C_, 32>), grid, dim3(kGThreads), tile_bytes, stream, a); } while (0)
#else
#define JA_GE(C_) hipLaunchKernelGGL((k_generic_encode<C_, 32>), grid, dim3(kGThreads), tile_bytes, stream, a)
#endif
#endif
    switch (L.nplanes) {
    case 1: JA_GE(1); break;
    case 2: JA_GE(2); break;
    case 3: JA_GE(3); break;
    default: hipLaunchKernelGGL((k_generic_encode<4, 32>), grid, dim3(kGThreads), tile_bytes, stream, a); break;
    }
#undef JA_GE
    return hipGetLastError();
}

}  // namespace jpeg_amd
