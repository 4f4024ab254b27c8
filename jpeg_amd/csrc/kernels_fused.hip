// kernels_fused.hip -- fused Spectral -> pixels fast path for the built-in 8-bit formats other than 4:2:0.
//
// Replaces idct() -> interleaved(cosite: false) -> unpack(as:) (decode.swift:4154, 4182,
// 4294) for y8 images and for ycc8 images whose luma has the full sampling factor and whose chroma planes are
// subsampled 1x or 2x on ONE axis at most (4:4:4, 4:2:2, 4:4:0), without materialising Planar / Rectangular -- or
// anything else -- in HBM: one launch of k_luma_fused.  (4:2:0 couples strips vertically AND horizontally: k_quad420,
// kernels_quad.hip.)
//
//   k_luma_fused    one luma 8x8 block per work-item: dequantise + IDCT in registers; the wave transforms the chroma
//                   blocks under its strip itself -- 4:4:4: every work-item its own Cb and Cr block; 4:2:2 and 4:4:0:
//                   one pass for the 64 blocks under the strip and one for the neighbour blocks whose edge column
//                   (4:2:2) or edge row (4:4:0) the filter reaches into -- and keeps their samples as bytes in a
//                   wave-private LDS tile (+1 sample halo, clamped to the padded plane like decode.swift:4245-4246);
//                   bilinear upsample, YCbCr->RGB, pack and store 8 rows x 24 B.
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
// Development switches (never defined in the product build; tools/build_exp.sh makes A/B builds):
// JA_X_NOIDCT / JA_X_NOCOLOR / JA_X_NOSTORE (the kernel without its transform / without its upsampling and colour
// arithmetic / without its stores: tools/ablate.sh), JA_X_NOPRIO.
#pragma clang fp contract(off)

#include "dct.hpp"
#include "fused_common.hpp"
#include "kernels.hpp"
#include "upsample.hpp"

#include <algorithm>
#include <cstdlib>

namespace jpeg_amd {

namespace {

// a strip is BX x BY luma blocks, BX * BY == 64 (one block per work-item): 32 x 2, or 16 x 4 for
// images whose width leaves the last 32-block strip half empty (1920 px = 7.5 strips of 32)

// ---------------------------------------------------------------------------------------
// luma IDCT + chroma IDCT + upsample + colour + store
// ---------------------------------------------------------------------------------------
struct LumaArgs {
    const int16_t *coef;
    size_t coef_stride;
    const int16_t *ccoef[2]; // the chroma coefficient planes
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
    int pair;                      // 4:4:0: a unit of the walk is a PAIR of strip rows (calls with more strips than resident waves)
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
// Waves per SIMD a variant is built for: what its LDS footprint admits (three workgroups of ~51 KiB per CU for grey; the
// layouts with a full-width or full-height chroma tile and 4:4:4 need 58-75 KiB per workgroup: two).
template <int SX, int SY, bool CHROMA>
constexpr int luma_waves_per_simd() { return (CHROMA && SX != SY) ? 2 : 3; }

template <int SX, int SY, int MODE, bool CHROMA, bool FAST, int BX>
__global__ __launch_bounds__(kThreads, (luma_waves_per_simd<SX, SY, CHROMA>())) void k_luma_fused(LumaArgs a)
{
    static_assert(!(CHROMA && SX == 2 && SY == 2), "4:2:0 is decoded by k_quad420 (kernels_quad.hip)");
    constexpr int BY = 64 / BX;                          // block rows per strip
    constexpr int NW = kThreads / 64;                    // waves per workgroup
    constexpr int CW = BX * 8 / SX;                      // chroma samples per strip row
    constexpr int CR = BY * 8 / SY;                      // chroma rows under a strip
    constexpr int HX = SX == 2 ? 4 : 0;                  // halo bytes per side
    constexpr int HY = SY == 2 ? 1 : 0;
    constexpr int PITCH = (CW + 2 * HX) / 4;             // dwords per LDS row
    // 4:4:0 (round 6): a wave's unit of work is a PAIR of vertically adjacent strips that share one chroma tile -- the sample row
    // between them is the other strip's own first / last row, so only the pair's outer block rows need the edge-row pass:
    // 2 + 1/3 transform passes per strip become 2 + 1/6.  No synchronisation: both strips belong to the same wave.
    constexpr bool PAIR = CHROMA && SX == 1 && SY == 2;
    constexpr int ROWS = (PAIR ? 2 : 1) * CR + 2 * HY;
    // 4:4:4: every work-item transforms the Cb and Cr blocks that lie under its luma block itself (same geometry),
    // parks their samples as bytes in 32 registers and then does the luma block
    constexpr bool INTHREAD = CHROMA && SX == 1 && SY == 1;
    constexpr int PLANE = (CHROMA && !INTHREAD) ? ROWS * PITCH : 1;   // dwords per plane of the tile
    constexpr int SEG_DW = BX * 6;                       // one pixel row of one block row: 24 B per block
    constexpr int CPS = SEG_DW / 4;                      // 16-byte chunks per such segment
    // 4:2:2 (wide strips): the 16 x 2 chroma blocks per plane under a strip are exactly one block per
    // work-item for both planes together; a second, nearly empty pass transforms the 8 neighbour
    // blocks that supply the one-sample halo left and right.  No chroma intermediate in HBM (round 2; it was
    // 134 of 604 MB at 8192 x 8192).
    constexpr bool IN422 = CHROMA && SX == 2 && SY == 1 && BX == 32;
    // 4:4:0 (narrow strips, 16 x 4 luma blocks): the 16 x 2 chroma blocks per plane under a strip are again one block per
    // work-item, and the 2 x 16 x 2 blocks of the block rows above and below -- of which only the nearest sample row is
    // wanted (idct_block_edge_row: a third of a block's arithmetic) -- one more.  No chroma intermediate in HBM (round 3:
    // a first launch used to write the Cb / Cr samples as bytes, 134 of 604 MB at 8192 x 8192).
    constexpr bool IN440 = CHROMA && SX == 1 && SY == 2 && BX == 16;
    static_assert(!(CHROMA && SX == 1 && SY == 2) || IN440, "4:4:0 is written for the 16 x 4 strip");
    constexpr bool INSTRIP = INTHREAD || IN422 || IN440;   // the kernel transforms its chroma blocks itself
    constexpr int NTAB = INSTRIP ? 3 : 1;
    __shared__ __attribute__((aligned(16))) uint32_t coefbuf[NW][64 * 32];  // 8 KiB per wave
    __shared__ __attribute__((aligned(16))) uint32_t stage[NW][BY * SEG_DW]; // one pixel row x BY block rows
    __shared__ uint32_t scw[NW][INTHREAD ? 1 : 2 * PLANE];  // chroma samples under the strip (+ halo); 4:4:4 keeps its own in registers
    __shared__ __attribute__((aligned(16))) float sqw[NW][NTAB][64];   // modulated table(s): luma (, Cb, Cr) -- TRANSPOSED ([8 k + h], dct.hpp TransposedTable)

    const int lane0 = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // strip math stays scalar
    uint32_t *stage_w = stage[wave];
    uint32_t *coef_w = coefbuf[wave];
    // LDS byte address of the wave's coefficient buffer (low 32 bits of the flat shared address)
    const uint32_t coef_lds = lds_address(coef_w);
    uint32_t *sc = scw[wave];
    float *sq = sqw[wave][0];

    // strip s -> image, strip row (BY block rows), strip column (BX blocks)
    FastDiv fd_tpi, fd_tx;   // by strips per image, strips per row
    fd_tpi.set((uint32_t)a.tiles_per_image); fd_tx.set((uint32_t)a.tiles_x);
    auto locate = [&](int s, int &img, int &syi, int &sxi) {
        uint32_t rem, col;
        img = (int)fd_tpi.div((uint32_t)s, rem);
        syi = (int)fd_tx.div(rem, col);
        sxi = (int)col;
    };

    // LDS-DMA of the 64 blocks of strip s: an instruction moves 64 x 16 B; slot u = 64 i + lane holds chunk
    // (u & 7) ^ ((b >> 1) & 7) of block b = u >> 3 (the XOR on the SOURCE address makes the later per-work-item ds_read_b128
    // conflict-free).  Round 5: runs of neighbouring blocks go through BUFFER RESOURCES over the block rows in question
    // (base advanced to the first row in 64 bits -- a plane may exceed 4 GiB -- num_records = those rows inside the plane):
    // one scalar byte offset per run; rows outside the plane and runs that overhang the resource arrive as zeros, runs that
    // overhang a block row fetch the head of the next one -- "fetched, not used" either way (their pixels are dropped by the
    // store's range check, their chroma samples repaired in the tile).  No clamped addresses, no interior / edge split.
    // which: 0 luma; 4:4:4: 1 Cb, 2 Cr; 4:2:2: 1 both chroma planes under the strip, 2 their halo blocks (block by block);
    // 4:4:0: 1 the chroma blocks under the strip, 3 the block rows above and below
    auto sgpr = [](uint32_t v) -> uint32_t { asm volatile("" : "+s"(v)); return v; };   // (see kernels_quad.hip)
    auto rows_of = [&](const int16_t *plane, uint32_t row0, uint32_t nrows, uint32_t row_blocks) -> i32x4_t {
        return make_srd(reinterpret_cast<const char *>(plane) + ((uint64_t)(row0 * row_blocks) << 7), (nrows * row_blocks) << 7);
    };
    // (the strip at strip row `syi`, strip column `sxi` of image `img`; `rows`, which == 3 only: the pair's strip rows, 1 or 2)
    auto dma_at = [&](int img, int syi, int sxi, int lane, int which = 0, int rows = 1) {
        const uint32_t l3 = lane >> 3;
        const uint32_t ve = l3 * 128 + (((lane & 7) ^ (l3 >> 1)) << 4);  // even pieces; odd pieces: chunk ^ 4
        if constexpr (IN422 || IN440) {
            const uint32_t uxc = sgpr((uint32_t)(a.pw_c >> 3));
            const int uyc = a.ph_c >> 3;
            if (which == 1) {   // block b of the buffer: plane b >> 5, row (b >> 4) & 1, column b & 15: four runs of sixteen
#pragma unroll
                for (int pl = 0; pl < 2; ++pl) {
                    const i32x4_t srd = rows_of(a.ccoef[pl] + img * a.ccoef_stride[pl], (uint32_t)(2 * syi), (uint32_t)min(2, uyc - 2 * syi), uxc);
#pragma unroll
                    for (int r = 0; r < 2; ++r)
                        lds_dma16_brun<2, true>(srd, ((uint32_t)r * uxc + (uint32_t)(16 * sxi)) << 7, ve, ve ^ 64u, coef_lds + 2048 * (2 * pl + r));
                }
                return;
            }
            if (which == 3) {   // 4:4:0 halo: block b of the buffer: below b >> 5, plane (b >> 4) & 1, column b & 15
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int row = (u >> 1) ? 2 * syi + 2 * rows : 2 * syi - 1;   // below the unit's last strip / above its first
                    const bool there = row >= 0 && row < uyc;   // a missing row: an empty resource (zeros; the tile row is repaired below)
                    const i32x4_t srd = rows_of(a.ccoef[u & 1] + img * a.ccoef_stride[u & 1], there ? (uint32_t)row : 0u, there ? 1u : 0u, uxc);
                    lds_dma16_brun<2, false>(srd, (uint32_t)(16 * sxi) << 7, ve, ve ^ 64u, coef_lds + 2048 * u);
                }
                return;
            }
            if (which == 2) {   // block b = 0..7: plane b >> 2, side (b >> 1) & 1 (0 left, 1 right), row b & 1
                const int b = lane >> 3;
                const int16_t *cbase = a.ccoef[b >> 2] + img * a.ccoef_stride[b >> 2];
                const int bx = ((b >> 1) & 1) ? 16 * sxi + 16 : 16 * sxi - 1, by = 2 * syi + (b & 1);
                const uint32_t blk = (bx >= 0 && bx < (int)uxc && by < uyc) ? (uint32_t)by * uxc + bx : 0u;
                const int c = (lane & 7) ^ ((b >> 1) & 7);
                lds_dma16(reinterpret_cast<const char *>(cbase) + ((size_t)blk * 128 + 16 * c), coef_lds);
                return;
            }
        }
        const int16_t *base = a.coef + img * a.coef_stride;
        if constexpr (INTHREAD) {
            if (which) base = a.ccoef[which - 1] + img * a.ccoef_stride[which - 1];
        }
        // BY runs of BX neighbouring blocks of the strip's block rows
        const uint32_t ux = sgpr((uint32_t)a.ux);
        const i32x4_t srd = rows_of(base, (uint32_t)(BY * syi), (uint32_t)min(BY, a.uy - BY * syi), ux);
#pragma unroll
        for (int r = 0; r < BY; ++r)
            lds_dma16_brun<BX / 8, true>(srd, ((uint32_t)r * ux + (uint32_t)(sxi * BX)) << 7, ve, ve ^ 64u, coef_lds + r * (BX * 128));
    };

    // 4:2:2, second chroma pass ((block, column) work-items): where the eight coefficients of this lane's column lie in the
    // DMA image of the 8 halo blocks (slot 8 b + i holds chunk i ^ ((b >> 1) & 7) of block b), fixed for the whole walk
    uint32_t edge_off[8] = {};
    int edge_k = 0;
    if constexpr (IN422) {
        const int b = lane0 >> 3;
        edge_k = edge_col_of_lane(lane0 & 7);
#pragma unroll
        for (int hh = 0; hh < 8; ++hh) {
            const int zz = zigzag_of(edge_k, hh);
            edge_off[hh] = 16u * (8u * b + (uint32_t)((zz >> 3) ^ ((b >> 1) & 7))) + 2u * (zz & 7);
        }
    }

    // The wave's walk: strips first_tile + k, k = wave index, wave index + resident waves, ...
    const int nwaves = (int)gridDim.x * NW;   // the walk's stride
    FastDiv fd_nw; fd_nw.set((uint32_t)nwaves);
    const int len = a.total_tiles - a.first_tile;
    auto valid = [&](int k) -> bool { return k < len; };
    auto strip_at = [&](int k) -> int { return a.first_tile + k; };
    int k = (int)blockIdx.x * NW + wave;
    if (k >= len) return;
    int s = strip_at(k);
    // a unit is a strip -- or, 4:4:0, a pair of strip rows (locate() then yields the PAIR row): its strips are walked in turn
    const int strips_y = (a.uy + BY - 1) / BY;
    auto dma_unit_head = [&](int unit, int lane) {   // what a unit needs first: its (first strip's) chroma blocks, or its luma blocks
        int img, uyi, sxi;
        locate(unit, img, uyi, sxi);
        dma_at(img, (PAIR && a.pair) ? 2 * uyi : uyi, sxi, lane, INSTRIP ? 1 : 0);
    };
    dma_unit_head(s, lane0);
    int img_of_table = -1;
    int stores_behind_dma = 0;  // wave-uniform
    for (; k < len; k += nwaves, s = strip_at(min(k, len - 1))) {
      int img, uyi, sxi;
      locate(s, img, uyi, sxi);
      const bool paired = PAIR && a.pair;   // (wave-uniform; small calls walk single strips: a pair is two strips in a row on ONE wave)
      const int syi0 = paired ? 2 * uyi : uyi;
      const int nph = (paired && syi0 + 1 < strips_y) ? 2 : 1;   // strips of this unit
#pragma unroll 1
      for (int ph = 0; ph < nph; ++ph) {
        // Launder the lane id once per strip: everything below that depends only on the lane is
        // cheap to recompute, but hoisted out of this loop it would pin ~60 VGPRs for good.
        int lane = lane0;
        asm volatile("" : "+v"(lane));
        const int lbx = lane & (BX - 1), seg = (int)((unsigned)lane / BX);
        const int syi = syi0 + ph;

        // ---- modulated table (only when the image changes) ----
        if (img != img_of_table) {
            const int qk = lane & 7, qh = lane >> 3;
            sq[8 * qk + qh] = modulate_entry(qk, qh, 0.125f, a.quanta[img * a.quanta_stride + 64 * a.qi + zigzag_of(qk, qh)]);
            if constexpr (INSTRIP) {
#pragma unroll
                for (int pl = 0; pl < 2; ++pl)
                    sqw[wave][1 + pl][8 * qk + qh] = modulate_entry(qk, qh, 0.125f,
                        a.quanta[img * a.quanta_stride + 64 * a.cqi[pl] + zigzag_of(qk, qh)]);
            }
            img_of_table = img;
        }

        // ---- this strip's coefficients: wait for the DMA, read 8 x 16 B (swizzled).  VM
        //      operations retire in issue order and the DMA was issued BEFORE the previous
        //      strip's pixel stores: when that strip took the branch-free store path (exactly
        //      2 store instructions per pixel row) only the DMA has to be waited for, not the
        //      16 stores behind it. ----
        if (stores_behind_dma == 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#ifndef JA_X_NOPRIO
        // The waves of a SIMD do not advance at the same pace: the scheduler issues the oldest ready wave first, so with
        // equal shares the first wave of a SIMD is done long before the last (8192 x 8192, four strips each: ends between
        // 30 and 72 us, tools/phase_profile.py) and the SIMD spends the end of the launch with one or two waves -- too
        // few to keep it busy.  A wave with more strips left therefore runs at a higher priority: the laggards catch up
        // and all waves of a SIMD leave within a strip of each other (ends between 47 and 66 us).
        {
            uint32_t rr_;
            const int rem = (int)fd_nw.div((uint32_t)(len - 1 - k), rr_);   // strips after this one
            if (rem >= 3) __builtin_amdgcn_s_setprio(3);
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
        read_block();
        // 4:4:4: the work-item's own Cb / Cr samples, four to a dword, wait in 32 registers while the luma block is
        // transformed (as 8 KiB of LDS per wave they held the kernel at two waves per SIMD)
        uint32_t stash[2][INTHREAD ? 16 : 1];
        if constexpr (INTHREAD) {
            // Cb, then Cr: while one plane is transformed the next one's coefficients are on their
            // way into the (single) LDS buffer -- the block has to be in registers before the DMA
            // may overwrite it, hence the lgkmcnt wait
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                dma_at(img, syi, sxi, lane, pl == 0 ? 2 : 0);
                float g[64];
                idct_block(w, TransposedTable{sqw[wave][1 + pl]}, 128.5f, g);
                {
                    uint32_t pk[16];   // clamp [0, 255] + truncate (trunc_pack*, fused_common.hpp)
                    trunc_pack24(g, pk); trunc_pack24(g + 24, pk + 6); trunc_pack16(g + 48, pk + 12);
#pragma unroll
                    for (int i = 0; i < 16; ++i) stash[pl][i % (INTHREAD ? 16 : 1)] = pk[i];
                }
#pragma unroll
                for (int i = 0; i < (INTHREAD ? 16 : 1); ++i) asm volatile("" : "+v"(stash[pl][i]));
                __builtin_amdgcn_sched_barrier(0);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                read_block();
            }
        }

        if constexpr (IN422) {
            // pass 1: the strip's own chroma blocks (lane: plane, block row, block column)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            dma_at(img, syi, sxi, lane, 2);
            {
                const int pl = lane >> 5, r = (lane >> 4) & 1, c = lane & 15;
                float g[64];
                idct_block(w, TransposedTable{sqw[wave][1 + pl]}, 128.5f, g);
                uint32_t *dst = sc + pl * PLANE + 8 * r * PITCH + 1 + 2 * c;
                uint32_t pk[16];   // clamp [0, 255] + truncate (trunc_pack*, fused_common.hpp)
                trunc_pack24(g, pk); trunc_pack24(g + 24, pk + 6); trunc_pack16(g + 48, pk + 12);
#pragma unroll
                for (int y = 0; y < 8; ++y) { dst[y * PITCH] = pk[2 * y]; dst[y * PITCH + 1] = pk[2 * y + 1]; }
            }
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            // pass 2: the 8 neighbour blocks' edge columns -> the tile's halo dwords.  One (block, column) per work-item
            // (idct_edge_col_split, dct.hpp): group b = lane >> 3 is block b of the buffer -- plane b >> 2, side (b >> 1) & 1
            // (0 left, 1 right), block row b & 1 -- and lane j of the group its first-pass column edge_col_of_lane(j); the
            // coefficients are read straight from the DMA image (16-bit LDS reads at offsets fixed before the walk).
            {
                const int b = lane >> 3, pl = b >> 2, side = (b >> 1) & 1, r = b & 1;
                int cf[8];
                float qv[8];
                const char *cimg = reinterpret_cast<const char *>(coef_w);
                const float *qcol = sqw[wave][1 + pl] + 8 * edge_k;   // transposed table: the column's eight entries are contiguous
#pragma unroll
                for (int hh = 0; hh < 8; ++hh) {
                    cf[hh] = *reinterpret_cast<const int16_t *>(cimg + edge_off[hh]);
                    qv[hh] = qcol[hh];
                }
                // the block is in registers: the strip's luma blocks may follow it into the buffer
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                dma_at(img, syi, sxi, lane, 0);
                float edge[8];
                idct_edge_col_split(cf, qv, 128.5f, side != 0, lane & 7, edge);
                uint32_t e[8];
                trunc_bytes8(edge, e);
                const bool exists = side ? 16 * sxi + 16 < (a.pw_c >> 3) : sxi > 0;
                if ((lane & 7) == 0 && exists) {
                    uint32_t *dst = sc + pl * PLANE + 8 * r * PITCH + (side ? PITCH - 1 : 0);
#pragma unroll
                    for (int y = 0; y < 8; ++y) dst[y * PITCH] = e[y] * 0x01010101u;
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

        if constexpr (IN440) {
          if (ph == 0) {   // the unit's first strip: the chroma tile of the whole unit (the second strip finds it ready)
            const int uyc = a.ph_c >> 3;
            // the strips' own chroma blocks (lane: plane, block row, block column) -> tile rows 1 .. 16 (first strip), 17 .. 32 (second)
#pragma unroll 1
            for (int st = 0; st < nph; ++st) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                if (st + 1 < nph) dma_at(img, syi0 + 1, sxi, lane, 1);      // the second strip's chroma blocks ...
                else dma_at(img, syi0, sxi, lane, 3, nph);                   // ... or the block rows above and below the unit
                {
                    const int pl = lane >> 5, r = (lane >> 4) & 1, c = lane & 15;
                    float g[64];
                    idct_block(w, TransposedTable{sqw[wave][1 + pl]}, 128.5f, g);
                    uint32_t *dst = sc + pl * PLANE + (HY + CR * st + 8 * r) * PITCH + 2 * c;
                    uint32_t pk[16];   // clamp [0, 255] + truncate (trunc_pack*, fused_common.hpp)
                    trunc_pack24(g, pk); trunc_pack24(g + 24, pk + 6); trunc_pack16(g + 48, pk + 12);
#pragma unroll
                    for (int y = 0; y < 8; ++y) { dst[y * PITCH] = pk[2 * y]; dst[y * PITCH + 1] = pk[2 * y + 1]; }
                }
                __builtin_amdgcn_sched_barrier(0);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                read_block();
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            dma_at(img, syi0, sxi, lane, 0);
            // the block rows above and below the unit (lane: below, plane, column): the last / first sample row of each
            // -> tile rows 0 and CR nph + 1
            {
                const bool below = lane >= 32;
                const int pl = (lane >> 4) & 1, c = lane & 15;
                float r8[8];
                idct_block_edge_row(w, TransposedTable{sqw[wave][1 + pl]}, 128.5f, (uint32_t)(lane - 32) & 0x80000000u, r8);   // above (lane < 32): the last row
                uint32_t p01[2];
                trunc_pack8(r8, p01);
                if (below ? 2 * syi0 + 2 * nph < uyc : syi0 > 0) {
                    uint32_t *dst = sc + pl * PLANE + (below ? HY + CR * nph : 0) * PITCH + 2 * c;
                    dst[0] = p01[0]; dst[1] = p01[1];
                }
            }
            // the plane's top and bottom: the reference clamps the sample row (decode.swift:4246) -- a missing row is the
            // nearest own row.  Only this wave reads its tile.
            {
                const int rows_avail = a.ph_c - syi0 * CR;   // sample rows of the plane from the first one under the unit
                auto copy_row = [&](int dstr, int srcr) {
                    for (int d = lane; d < 2 * PITCH; d += 64) {
                        uint32_t *col = sc + (d >= PITCH ? PLANE + d - PITCH : d);
                        col[dstr * PITCH] = col[srcr * PITCH];
                    }
                };
                if (syi0 == 0) copy_row(0, 1);
                if (rows_avail > 0 && rows_avail <= CR * nph) copy_row(rows_avail + 1, rows_avail);
            }
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            read_block();
          }
        }

        // ---- luma: dequantise + IDCT, clamp + truncate (decode.swift:4121-4122), kept as
        //      integer-valued floats for the colour matrix ----
        float yv[64];
#ifdef JA_X_NOIDCT  // experiment: how long is a strip without the IDCT arithmetic?
#pragma unroll
        for (int i = 0; i < 64; ++i) yv[i] = (float)(w[i & 31] >> (i & 32 ? 16 : 0) & 0xff);
#else
        idct_block(w, TransposedTable{sq}, 128.5f, yv);
#pragma unroll
        for (int i = 0; i < 64; ++i) yv[i] = floorf(__builtin_amdgcn_fmed3f(yv[i], 0.0f, 255.0f));
#endif

        // Pin the IDCT HERE: LLVM otherwise sinks it below the waits / DMA (its results are first
        // used in the colour phase) and the wave would park on the chroma rows before doing any
        // arithmetic instead of letting them land during the IDCT.
#pragma unroll
        for (int i = 0; i < 64; ++i) asm volatile("" : "+v"(yv[i]));
        __builtin_amdgcn_sched_barrier(0);

        // ---- the coefficient buffer is consumed: prefetch the next strip into it.  From here to
        //      the end of the strip only stores are issued, so nothing waits on the DMA. ----
        if (ph + 1 < nph) dma_at(img, syi + 1, sxi, lane, 0);                     // the unit's second strip: its luma blocks
        else if (valid(k + nwaves)) dma_unit_head(strip_at(k + nwaves), lane);   // the next unit
        // keep the phases apart (hoisting the chroma LDS reads above the IDCT costs ~70 VGPRs)
        __builtin_amdgcn_sched_barrier(0);

        // ---- chroma rows, produced just in time from the LDS tile ----
        // Chroma row j of this block's patch: the LDS reads (hraw) and the conversion + horizontal
        // interpolation, x4 when SX == 2 (hconv), are separate so that the reads can be issued one pixel row
        // ahead of their use -- a wave that waits ~150 cycles for LDS ten times per strip leaves its SIMD to
        // two other waves that are as likely to be waiting themselves.
        auto hraw = [&](int pl, int j, uint32_t (&r)[3]) {
            const uint32_t *row = sc + pl * PLANE + ((PAIR ? ph * CR : 0) + seg * (8 / SY) + j) * PITCH;
            if constexpr (SX == 2) {
                r[0] = row[HX / 4 - 1 + lbx]; r[1] = row[HX / 4 + lbx]; r[2] = row[HX / 4 + 1 + lbx];
            } else if constexpr (INTHREAD) {   // the block's own samples, parked above
                r[0] = stash[pl][(2 * j) % (INTHREAD ? 16 : 1)]; r[1] = stash[pl][(2 * j + 1) % (INTHREAD ? 16 : 1)]; r[2] = 0;
            } else {
                r[0] = row[2 * lbx]; r[1] = row[2 * lbx + 1]; r[2] = 0;
            }
        };
        // 4:2:2 / 4:4:0: the samples enter as 2^15 + p + 1/32 (ubyte_magic, upsample.hpp) so that the ONE 3a + b step is exact
        // (2^17 + v + 1/8) and its rounding needs no floor (finish)
        auto hconv = [&](const uint32_t (&r)[3], float (&o)[8]) {
            if constexpr (SX == 2) {
                const float p[6] = {ubyte_magic<3>(r[0]), ubyte_magic<0>(r[1]), ubyte_magic<1>(r[1]),
                                    ubyte_magic<2>(r[1]), ubyte_magic<3>(r[1]), ubyte_magic<0>(r[2])};
                lerp_row_2x(p, o);
            } else if constexpr (SY == 2) {
                o[0] = ubyte_magic<0>(r[0]); o[1] = ubyte_magic<1>(r[0]); o[2] = ubyte_magic<2>(r[0]); o[3] = ubyte_magic<3>(r[0]);
                o[4] = ubyte_magic<0>(r[1]); o[5] = ubyte_magic<1>(r[1]); o[6] = ubyte_magic<2>(r[1]); o[7] = ubyte_magic<3>(r[1]);
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
        // final chroma value of one pixel.  4:4:4: the sample itself [- 128].  One subsampled axis: floor(v / 4 + 1/2) [- 128] of
        // v = 3a + b, without a floor -- V = 2^17 + v + 1/8 arrives exact, and fma(V, 1/4, 1.5 * 2^23 - 2^15 [- 128]) is ONE rounding of
        // 1.5 * 2^23 [- 128] + v / 4 + 1/32 to a float whose ulp is 1; v / 4 + 1/32 is never half-way, and at v / 4 = n + 1/2 the 1/32
        // tips it up like the reference's round-half-away (tests/test_colour_rounding.py enumerates every byte pair).  An FMA and a
        // subtraction instead of an FMA and a v_floor_f32 (half rate): -128 slow instructions per strip.
        auto finish = [&](float v) -> float {
            if constexpr (SX == 1 && SY == 1) return MODE == 1 ? v - 128.0f : v;
            else return __builtin_fmaf(v, 0.25f, kMagic - 32768.0f - (MODE == 1 ? 128.0f : 0.0f)) - kMagic;
        };

        float hw[2][3][8];  // SY == 2: patch rows j-1, j, j+1 of both planes (sliding window)
        uint32_t rawn[2][3] = {{0, 0, 0}, {0, 0, 0}};   // LDS dwords of the patch row that is converted next
        if constexpr (CHROMA && SY == 2) {
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
        if constexpr (BX == 32) { sg0 = 1 + ((lane - 48) >> 31); sg1 = 1; asm volatile("" : "+v"(sg0)); }   // (lane >= 48 ? 1 : 0, without a select)
        else { sg0 = (int)((unsigned)lane / CPS); sg1 = (int)((64u + (unsigned)lane) / CPS); }
        const int j0 = lane - CPS * sg0, j1 = 64 + lane - CPS * sg1;
        // inside the image: 16 j < nb (and lane < 32 for the second chunk) as the SIGN of a difference (kernels_quad.hip: no v_cndmask_b32)
        const int in0 = 16 * j0 - nb, in1 = max(16 * j1 - nb, lane - 32);            // negative: inside
        const bool col0 = in0 < 0, col1 = in1 < 0;
        // The strip's rows through a BUFFER RESOURCE (round 5, as in k_quad420): base = the strip's first pixel, num_records = the
        // bytes to the end of its last row INSIDE the image -- a row below the image is out of range and the hardware drops its
        // store; a chunk right of the image gets an out-of-range voffset.  Two store instructions per pixel row, no predicate, no
        // branch, the row is the scalar soffset.  FAST = false (any width): whole chunks the same way, the partial last chunk of
        // a row (right-edge column only) dword- and byte-wise (store_tail).
        const int rows_here = min(8 * BY, a.H - 8 * BY * syi);
        const uint32_t strip_bytes = (uint32_t)rows_here * pitch;
        const i32x4_t out_srd = make_srd(strip_out, strip_bytes);
        const int rem0 = col0 ? nb - 16 * j0 : 0, rem1 = col1 ? nb - 16 * j1 : 0;   // bytes of the lane's chunks inside the image
        const uint32_t base0 = sg0 * 8u * pitch + 16u * j0, base1 = sg1 * 8u * pitch + 16u * j1;
        auto place = [](uint32_t base, int d) -> uint32_t {      // base where d < 0, an out-of-range voffset elsewhere: one v_bfi_b32
            const uint32_t m = (uint32_t)(d >> 31);
            return (base & m) | (0x80000000u & ~m);
        };
        const uint32_t voff0 = place(base0, FAST ? in0 : in0 + 15), voff1 = place(base1, FAST ? in1 : max(16 * j1 - nb + 15, lane - 32));
        stores_behind_dma = FAST ? 16 : 0;

        // One pixel row of the strip's BY block rows at a time.  The row's LDS and memory traffic is software-
        // pipelined behind the NEXT row's arithmetic: row y is staged (ds_write) and read back as 16-byte chunks
        // (ds_read) right after its arithmetic, but the chunks are stored only after the arithmetic of row y + 1;
        // the chroma dwords of the next patch row are requested a row (SY == 1) or two (SY == 2) ahead.
        uint4 pv0 = make_uint4(0, 0, 0, 0), pv1 = make_uint4(0, 0, 0, 0);   // chunks of the previous pixel row
        auto store_tail = [&](const uint4 &v, uint32_t base, int rem, uint32_t soff) {   // the first `rem` (1 .. 15) bytes of a chunk
            const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(strip_out, 0, (int)strip_bytes, 0x00020000);
            const bool part = rem > 0 && rem < 16;
            const int nd = rem >> 2, nbytes = rem & 3;
            const uint32_t d[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int k = 0; k < 3; ++k)
                __builtin_amdgcn_raw_buffer_store_b32(d[k], rsrc, (part && k < nd) ? base + 4u * k : 0x80000000u, soff, 0);
            const uint32_t w = nd == 0 ? v.x : nd == 1 ? v.y : nd == 2 ? v.z : v.w;
#pragma unroll
            for (int b = 0; b < 3; ++b)
                __builtin_amdgcn_raw_buffer_store_b8((uint8_t)(w >> (8 * b)), rsrc, (part && b < nbytes) ? base + 4u * nd + b : 0x80000000u, soff, 0);
        };
        auto store_row = [&](int yy) {
#ifdef JA_X_NOSTORE  // experiment: everything but the global stores
            if (a.W < 0)
#endif
            {
                const u32x4_t q0 = {pv0.x, pv0.y, pv0.z, pv0.w}, q1 = {pv1.x, pv1.y, pv1.z, pv1.w};
                const uint32_t soff = (uint32_t)yy * pitch;   // scalar
                asm volatile("buffer_store_dwordx4 %0, %1, %4, %5 offen nt\n\t"
                             "buffer_store_dwordx4 %2, %3, %4, %5 offen nt\n\t"
                             "s_nop 0"   // a store of more than 64 bits with an SGPR offset: one wait state before its data registers may be rewritten
                             ::"v"(q0), "v"(voff0), "v"(q1), "v"(voff1), "s"(out_srd), "s"(soff) : "memory");
                if constexpr (!FAST) {
                    if (nb & 15) {   // wave-uniform: this strip column holds the image's right edge
                        store_tail(pv0, base0, rem0, soff);
                        store_tail(pv1, base1, rem1, soff);
                    }
                }
            }
        };
        // colour of pixel row y of the work-item's block from its luma samples and the row's chroma values; packed as 24 bytes
        auto colour_row = [&](int y, const float (&cv)[2][8], uint32_t (&d)[6]) {
            float c[24];
#pragma unroll
            for (int x = 0; x < 8; ++x) {
                const float yy = yv[8 * y + x];
                float c0, c1, c2;
                if constexpr (MODE == 1) {
                    if constexpr (CHROMA) {
                        const float pb = cv[0][x], pr = cv[1][x];
                        // jpeg.swift:441-453: x = (y + m_cb cb) + m_cr cr (the 0.0 * c terms are exact no-ops), clamped and
                        // TRUNCATED -- the pack below.  One FMA for R and B, two for G in this association (the other one is
                        // NOT exact): after the truncation every one of the 2^16 / 2^24 input combinations gives the
                        // reference's byte (tests/test_colour_rounding.py enumerates them).
                        c0 = __builtin_fmaf(1.40200f, pr, yy);
                        c1 = __builtin_fmaf(-0.71414f, pr, __builtin_fmaf(-0.34414f, pb, yy));
                        c2 = __builtin_fmaf(1.77200f, pb, yy);
                    } else {
                        c0 = c1 = c2 = yy;  // cb = cr = 128: every matrix term is +-0
                    }
                } else {
                    c0 = yy;
                    c1 = CHROMA ? cv[0][x] : 128.0f;
                    c2 = CHROMA ? cv[1][x] : 128.0f;
                }
                c[3 * x + 0] = c0; c[3 * x + 1] = c1; c[3 * x + 2] = c2;
            }
            trunc_pack24(c, d);   // clamp [0, 255] + truncate (fused_common.hpp)
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
            if constexpr (CHROMA) {
                if constexpr (SY == 2) {
                    if ((y & 1) == 1 && y < 7) {
#pragma unroll
                        for (int pl = 0; pl < 2; ++pl) hraw(pl, (y >> 1) + 3, rawn[pl]);
                    }
                } else if (y < 7) {
#pragma unroll
                    for (int pl = 0; pl < 2; ++pl) hraw(pl, y + 1, rawn[pl]);
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        store_row(7);
      }   // strips of the unit
    }
}



// Persistent grid = what is resident at once: workgroups per CU (LDS- and VGPR-bound, differs per
// instantiation: 3 for grey, 2 for the variants with a chroma tile or stash) x CUs.
template <int SX, int SY, int MODE, bool CHROMA, bool FAST, int BX>
int resident_workgroups()
{
    return resident_workgroups_of<k_luma_fused<SX, SY, MODE, CHROMA, FAST, BX>>(2);
}

template <int MODE, bool FAST, int BX>
hipError_t launch_luma(hipStream_t stream, int wgs, const LumaArgs &a, int sx, int sy, bool chroma)
{
    // grid = min(work, resident capacity)
    auto go = [&](auto kernel, int cap) {
        hipLaunchKernelGGL(kernel, dim3(wgs < cap ? wgs : cap), dim3(kThreads), 0, stream, a);
    };
#define JA_K(SX_, SY_, CH_) go(k_luma_fused<SX_, SY_, MODE, CH_, FAST, BX>, resident_workgroups<SX_, SY_, MODE, CH_, FAST, BX>());
    if (!chroma) JA_K(1, 1, false)
    else if (sx == 2 && sy == 1) { if constexpr (BX == 32) JA_K(2, 1, true) }
    else if (sx == 1 && sy == 2) { if constexpr (BX == 16) JA_K(1, 2, true) }
    else JA_K(1, 1, true)
#undef JA_K
    return hipGetLastError();
}

// Strip shape: 32 x 2 blocks unless 16 x 4 covers the plane with fewer strips (a half-empty strip
// costs as much as a full one).  Only the 4:4:4 / grey kernels come in both shapes: the 4:2:2 and 4:4:0 chroma
// tiles of a 16 x 4 strip would need 64 row transfers.
inline int strip_width(int ux, int uy, int sx, int sy)
{
    if (sx != sy) return sx == 2 ? 32 : 16;   // the in-strip chroma passes: 4:2:2 is written for the wide strip, 4:4:0 for the narrow one
    const long wide = (long)((ux + 31) / 32) * ((uy + 1) / 2), narrow = (long)((ux + 15) / 16) * ((uy + 3) / 4);
    return narrow < wide ? 16 : 32;
}

}  // namespace

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

// No layout keeps an intermediate in HBM any more (4:4:0 was the last, round 3): every kernel transforms the chroma blocks
// it needs itself.
hipError_t launch_fused_decode(hipStream_t stream, int n_images, const jpeg_amd_layout &L,
                               const PlaneSet &coef, QuantaRef q, bool rgb, uint32_t *d_walk_counters,
                               uint8_t *d_pixels, size_t pixel_stride)
{
    const bool chroma = L.nplanes == 3;
    if (chroma && L.scale_x == 2 && L.scale_y == 2)   // 4:2:0: the stack walk (kernels_quad.hip), one launch, no intermediate
        return launch_quad_decode(stream, n_images, L, coef, q, rgb, d_walk_counters, d_pixels, pixel_stride);
    // 4:4:4, 4:2:2, 4:4:0: k_luma_fused transforms the chroma blocks under (and around) its strips itself
    LumaArgs la{};
    if (chroma) {
        for (int i = 0; i < 2; ++i) {
            la.ccoef[i] = static_cast<const int16_t *>(coef.ptr[1 + i]);
            la.ccoef_stride[i] = coef.stride[1 + i];
            la.cqi[i] = L.qi[1 + i];
        }
        la.pw_c = 8 * L.units_x[1]; la.ph_c = 8 * L.units_y[1];
    }
    la.coef = static_cast<const int16_t *>(coef.ptr[0]);
    la.coef_stride = coef.stride[0];
    la.quanta = q.d_quanta; la.quanta_stride = q.image_stride; la.qi = L.qi[0];
    la.ux = L.units_x[0]; la.uy = L.units_y[0];
    la.W = L.width; la.H = L.height;
    la.out = d_pixels; la.out_stride = pixel_stride;
    // unit of work: strip of 32 x 2 (or 16 x 4) luma blocks; persistent waves
    const int sx = chroma ? L.scale_x : 1, sy = chroma ? L.scale_y : 1;
    const int bx = strip_width(la.ux, la.uy, sx, sy), by = 64 / bx;
    la.tiles_x = (la.ux + bx - 1) / bx;
    const int strips_y = (la.uy + by - 1) / by;
    // 4:4:0: a unit of the walk is a pair of strip rows (k_luma_fused, PAIR) -- when the call has more strips than waves are resident
    // (two workgroups of four per CU): a pair is two strips in a row on one wave, and a small call has waves to spare
    // (1920 x 1080: 10.8 us with single strips, 17.3 with pairs; 4096 x 4096: 30.9 against 26.7)
    const bool pairs = chroma && sx == 1 && sy == 2 && (long)la.tiles_x * strips_y * n_images > 4L * resident_workgroups<1, 2, 1, true, true, 16>();   // (four waves per workgroup)
    la.pair = pairs ? 1 : 0;
    la.tiles_per_image = la.tiles_x * (pairs ? (strips_y + 1) / 2 : strips_y);
    la.first_tile = 0;
    la.total_tiles = la.tiles_per_image * n_images;
    if (la.total_tiles == 0) return hipSuccess;
    const bool fast = (L.width & 15) == 0 && (pixel_stride & 15) == 0 &&
                      (reinterpret_cast<uintptr_t>(d_pixels) & 15) == 0;
    // grid == resident capacity (launch_luma clamps).  (Sizing it so that every wave gets the same number of
    // strips -- fewer waves, no thin last round -- measured 4 % slower; 2 instead of 3 workgroups per CU 3.5 %.)
    const int wgs = (la.total_tiles + 3) / 4;
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

}  // namespace jpeg_amd
