// kernels_quad.hip -- ycc8 4:2:0 Spectral -> YCbCr / RGB bytes in ONE launch, no intermediate in HBM (the "stack walk").
//
// Replaces idct() -> interleaved(cosite: false) -> unpack(as:) (decode.swift:4154, 4182, 4294) for three-plane images
// whose luma has the factors (2, 2) and whose chroma planes (1, 1): BASELINE.json configs[2] (8192 x 8192) and configs[4]
// (batches of 1920 x 1080).  6 B/px of algorithmic traffic: 128 B per coefficient block in, 3 B per pixel out.
//
// Work decomposition.  One 8 x 8 block per work-item (both IDCT passes and both transposes of decode.swift:3971-4099 are
// register renames).  A STRIP is BX x BY luma blocks with BX * BY = 64 -- 32 x 2 (256 x 16 px) or 16 x 4 (128 x 32 px) --
// and one wave's unit of luma work.  The bilinear chroma filter couples every pixel row to the chroma sample row above /
// below (decode.swift:4243-4257), so four vertically adjacent strips form a STACK (256 x 64 / 128 x 128 px) that the four
// waves of a workgroup decode together, sharing ONE chroma tile in LDS.  Per trip a wave runs
//   - its ROLE of the chroma pass: the stack's chroma blocks are dealt to the waves by kind -- two waves transform the 128
//     blocks under the stack (64 each), one the blocks above / below it (only one sample row of each is wanted: a third
//     of a block's arithmetic), one the neighbour blocks left / right (only the touching column: two thirds) -- and
//     writes its samples as bytes into the tile; the roles swap between the wave pairs every trip;
//   - the luma pass: dequantise + IDCT of its strip's 64 luma blocks;
//   - eight pixel rows: upsample from the tile (centred 2x: weights 1/4, 3/4), colour matrix, pack, store whole 16-byte
//     chunks of contiguous row segments through a small LDS staging row.
// Between the waves of a stack there is no barrier, only two monotonic LDS counters: a wave ARRIVES ("my samples are in
// the tile") right after its chroma role and checks the counter a luma transform later, before its first pixel row; the
// second counter keeps the tile from being overwritten while anyone still reads it.
// Coefficients arrive by LDS-DMA (global_load_lds_dwordx4) one phase ahead, into the wave's 8 KiB buffer: the luma blocks
// during the chroma transform, the NEXT stack's chroma blocks during the luma transform and the pixel rows.
//
// Exactness: dct.hpp op for op (-ffp-contract=off); upsample.hpp for the exact shortcuts of the filter (small-integer
// arithmetic, floor-free rounding); the colour matrix as in k_luma_fused (tests/test_colour_rounding.py enumerates every
// input).  Strips, stacks and tile columns that reach past the image or the plane are handled in place (clamped fetches,
// predicated stores, the reference's index clamps repaired in the tile), so any image size takes this kernel.
//
// Development switches (never defined in the product build): JA_PHASE_PROFILE, JA_X_NOSYNC, JA_X_NOCIDCT, JA_X_NOIDCT,
// JA_X_NOSTORE, JA_X_STAGGER=<cycles>, JA_X_GRID_PER_CU=<1 | 2>, JA_QUAD_WAVES=4 + JA_X_LDSHACK (round 4: what four waves per SIMD would be
// worth -- pairs of waves share a coefficient buffer, wrong pixels; profiles/r04_ab_four_waves_per_simd.txt).  (Round 3 also measured: the DMA instructions paced over the pixel rows or
// interleaved with the transform's columns, several priority schemes, progress feedback between the workgroups of a CU,
// one chroma pass per strip instead of the roles -- profiles/r03_ab_*.txt; those variants are in git history or under
// tools/exp_patches/, not in this file.)
#pragma clang fp contract(off)

#include "dct.hpp"
#include "fused_common.hpp"
#include "kernels.hpp"
#include "upsample.hpp"

#include <algorithm>
#include <cstdlib>

namespace jpeg_amd {

namespace {

JA_PHASE_STORAGE

struct QuadArgs {
    const int16_t *coef[3];        // Y, Cb, Cr coefficient planes [units_y][units_x][64], zigzag
    size_t coef_stride[3];         // elements between images
    const uint16_t *quanta;
    size_t quanta_stride;
    int qi[3];
    int ux, uy;                    // luma units
    int uxc, uyc;                  // chroma units
    int W, H;
    uint8_t *out;
    size_t out_stride;
    int tiles_x;                   // strip columns this launch walks ...
    int sx0;                       // ... starting at this one (units of BX blocks)
    uint32_t *tickets;             // dynamic walk (long calls): the number of stacks handed out beyond the first round (zeroed by
                                   // the host in front of the launch); nullptr: workgroup b walks stacks b, b + grid, b + 2 grid ...
    int stacks_per_image;          // tiles_x * stack rows
    int nstacks;                   // of the whole call
};

#ifndef JA_QUAD_WAVES
#define JA_QUAD_WAVES 3   // waves per SIMD the register allocation aims at
#endif
template <int BX> struct QuadShape {
    static constexpr int BY = 64 / BX;
    static constexpr int QS = kThreads / 64;              // strips (= waves) per stack: one stack per workgroup
};

// MODE: 0 = YCbCr bytes, 1 = RGB bytes.  FAST: W % 16 == 0 and 16-byte aligned rows, so every 16-byte chunk of a row
// segment is entirely inside the image or entirely outside (no tail code for the partial last chunk of a row).
template <int MODE, int BX, bool FAST, bool DYN>
__global__ __launch_bounds__(kThreads, JA_QUAD_WAVES) void k_quad420(QuadArgs a)
{
    constexpr int BY = QuadShape<BX>::BY, QS = QuadShape<BX>::QS;
    constexpr int NW = kThreads / 64;
    static_assert(QS == 4 && NW == 4, "the roles below are dealt to four waves");
    constexpr int CW = BX * 4, CR = BY * 4;               // chroma samples per strip row, chroma sample rows under a strip
    constexpr int CBW = BX / 2, CBR = BY / 2;             // chroma blocks under a strip: CBR rows of CBW
    constexpr int PITCH = CW / 4 + 2;                     // dwords per tile row: one halo dword left, the samples, one right
    constexpr int QROWS = QS * CR + 2;                    // sample rows of the stack's tile: halo, QS x CR rows, halo (34 / 66)
    constexpr int PLANE = QROWS * PITCH;                  // dwords per plane of the tile
    constexpr int SEG_DW = BX * 6;                        // one pixel row of one block row: 24 B per block
    constexpr int CPS = SEG_DW / 4;                       // 16-byte chunks per such segment
    // The stack's chroma blocks are dealt to its four waves BY KIND (the wave's ROLE of the trip), not by strip:
    //   role 0, 1   the 64 blocks under strips 0, 1 / 2, 3 of the stack (work-item b: strip b >> 5, plane (b >> 4) & 1, then
    //               row-major over CBR rows of CBW columns): a full transform, all work-items busy;
    //   role 2      the 4 CBW blocks above and below the stack (b / (2 CBW): above / below, plane, column), of which only
    //               the last / first sample row is wanted: idct_block_edge_row, a third of a block's arithmetic;
    //   role 3      the 4 (QS CBR + 2) neighbour blocks left and right (b >> 2: block row from the one above the stack to
    //               the one below, plane (b >> 1) & 1, side b & 1), of which only the column that touches the stack is
    //               wanted: idct_block_edge_cols, two thirds.
    // 3.0 transform passes per stack of four strips where one pass per strip (own blocks + a share of the halo in every
    // wave, round 2 / the first version of this kernel) is 4.0.  The roles swap between the wave pairs (0, 1) and (2, 3)
    // every trip, so that every wave does the same work over two trips.
    constexpr int NHROW = 4 * CBW, NSIDE = 4 * (QS * CBR + 2);   // blocks of role 2 / role 3: 64 / 24 (32 x 2 strips), 32 / 40
    constexpr int NDMA_C = 8;                             // LDS-DMA instructions of a chroma pass at most (8 blocks each)

#ifdef JA_X_LDSHACK   // experiment (wrong pixels): pairs of waves share a coefficient buffer -- what are four waves per SIMD worth?
    __shared__ __attribute__((aligned(16))) uint32_t coefbuf[NW / 2][64 * 32];
#else
    __shared__ __attribute__((aligned(16))) uint32_t coefbuf[NW][64 * 32];   // 8 KiB per wave
#endif
    __shared__ __attribute__((aligned(16))) uint32_t stage[NW][BY * SEG_DW]; // one pixel row x BY block rows
    __shared__ uint32_t qt[2 * PLANE];                    // the stack's tile; row 0: halo above, rows 1 + CR p ...: strip p, last row: halo below
    __shared__ __attribute__((aligned(16))) float sqw[NW][3][64];   // modulated tables: Y, Cb, Cr -- TRANSPOSED ([8 k + h], dct.hpp TransposedTable)
    // two monotonic counters.  [0] "ready": a wave has written its samples of this trip into the tile;
    // [1] "done": a wave has read the last sample of this trip.
    __shared__ uint32_t next_slot[2];                     // dynamic walk: the stack of the next trip (wave 0 publishes it, double-buffered)
    __shared__ uint32_t qsync[3];

    const int lane0 = threadIdx.x & 63;
    const int qp = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // the wave's strip of the stack; strip math stays scalar
#ifdef JA_X_LDSHACK
    uint32_t *coef_w = coefbuf[qp & 1], *stage_w = stage[qp];
#else
    uint32_t *coef_w = coefbuf[qp], *stage_w = stage[qp];
#endif
    const uint32_t coef_lds = lds_address(coef_w);
    uint32_t *sc = qt + CR * qp * PITCH;                  // this wave's window: row 0 = the sample row above its own rows
    uint32_t *ready = &qsync[0], *done = &qsync[1], *pub = &qsync[2];   // pub: dynamic walk, "the next stack is published"

    if (threadIdx.x < 3) qsync[threadIdx.x] = 0;
    __syncthreads();   // the only workgroup barrier of the walk

    FastDiv fd_spi, fd_tx;
    fd_spi.set((uint32_t)a.stacks_per_image); fd_tx.set((uint32_t)a.tiles_x);
    // stack -> image, strip row of this wave, strip column
    auto locate = [&](int q, int &img, int &syi, int &sxi) {
        uint32_t rem, col;
        img = (int)fd_spi.div((uint32_t)q, rem);
        syi = QS * (int)fd_tx.div(rem, col) + qp;
        sxi = (int)col + a.sx0;
    };
    const int strips_y = (a.uy + BY - 1) / BY;
#ifdef JA_X_ROLEROT   // experiment: every wave takes every role once in four trips (instead of swapping between the wave pairs)
    auto role_of = [&](int t) -> int { return (qp + t) & 3; };
#else
    auto role_of = [&](int t) -> int { return (qp + 2 * (t & 1)) & 3; };
#endif

    // LDS-DMA of a pass's blocks: an instruction moves 64 x 16 B; slot u = 64 i + lane of the wave's buffer holds chunk
    // (u & 7) ^ ((b >> 1) & 7) of block b = u >> 3 -- the XOR on the SOURCE address makes the later per-work-item ds_read_b128
    // (stride 128 B) bank-conflict-free.  The eight blocks of an instruction are neighbours in a block row: the block index is
    // scalar and only the lane's place inside the group is per lane (`ve`; odd pieces: chunk ^ 4).
    // Round 5: the runs go through BUFFER RESOURCES.  Luma: a resource over the strip's block rows inside the plane (base
    // advanced to the strip's first row in 64 bits -- a luma plane may exceed 4 GiB -- num_records = those rows); chroma: one
    // over the whole plane (at most 4096 x 4096 blocks = 2 GiB).  The run's place is one scalar byte offset.  Block rows above /
    // below the plane and runs that overhang the end of the resource are out of range and arrive as zeros; runs that overhang
    // the end of a block ROW fetch the head of the next row.  Either way those blocks are "fetched, not used": their pixels are
    // dropped by the store's range check, their chroma samples repaired in the tile (copy_row, fix_columns).  No clamped
    // addresses, no interior / edge split, no 64-bit address per run.
    auto lane_chunk = [&](int lane) -> uint32_t {
        const uint32_t l3 = lane >> 3;
        return l3 * 128 + (((lane & 7) ^ (l3 >> 1)) << 4);
    };
    // (sgpr(): the value re-defined in a scalar register.  Everything that ends in an "s" operand of the DMA statements is
    // computed from such copies: left alone, LLVM may keep a kernel argument that vector code also reads in a VGPR, select the
    // multiplications by it as VALU instructions and then have no scalar register to offer the statement)
    auto sgpr = [](uint32_t v) -> uint32_t { asm volatile("" : "+s"(v)); return v; };
    auto dma_luma = [&](int img, int syi, int sxi, int lane) {   // BY runs of BX blocks (the strip is not phantom)
        const uint32_t ve = lane_chunk(lane);
        const uint32_t ux = sgpr((uint32_t)a.ux);
        const uint32_t row0 = (uint32_t)(BY * syi), rows = (uint32_t)min(BY, a.uy - BY * syi);
        const char *base = reinterpret_cast<const char *>(a.coef[0] + img * a.coef_stride[0]) + ((uint64_t)(row0 * ux) << 7);
        const i32x4_t srd = make_srd(base, (rows * ux) << 7);
#pragma unroll
        for (int r = 0; r < BY; ++r)
            lds_dma16_brun<BX / 8, true>(srd, ((uint32_t)r * ux + (uint32_t)(sxi * BX)) << 7, ve, ve ^ 64u, coef_lds + r * (BX * 128));
    };
    // byte offset of block (row, col) of a chroma plane; a row above the plane (-1) wraps to an offset near 2^32 and a row
    // below it lies at or behind num_records (<= 2^31): both out of range
    auto dma_chroma = [&](int img, int syi, int sxi, int lane, int role) {
        const int top = syi - qp;                             // strip row of the stack's first strip
        const uint32_t uxc = sgpr((uint32_t)a.uxc);
        const uint32_t plane_bytes = sgpr(((uint32_t)a.uyc * uxc) << 7);
        auto chroma_off = [&](int row, int col) -> uint32_t { return ((uint32_t)row * uxc + (uint32_t)col) << 7; };
        if (role < 3) {   // runs of neighbouring blocks, one plane each
            const uint32_t ve = lane_chunk(lane);
            const char *base_b = reinterpret_cast<const char *>(a.coef[1] + img * a.coef_stride[1]);
            const char *base_r = reinterpret_cast<const char *>(a.coef[2] + img * a.coef_stride[2]);
            const i32x4_t srd_b = make_srd(base_b, plane_bytes), srd_r = make_srd(base_r, plane_bytes);
            if constexpr (BX == 32) {   // four runs of sixteen blocks
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int row = role < 2 ? top + 2 * role + (u >> 1) : ((u >> 1) ? CBR * (top + QS) : CBR * top - 1);
                    lds_dma16_brun<2, false>((u & 1) ? srd_r : srd_b, chroma_off(row, CBW * sxi), ve, ve ^ 64u, coef_lds + 2048 * u);
                }
            } else if (role < 2) {      // eight pieces of eight blocks, a block row each: plane (i >> 1) & 1
#pragma unroll
                for (int i = 0; i < NDMA_C; ++i)
                    lds_dma16_brun<1, false>(((i >> 1) & 1) ? srd_r : srd_b, chroma_off(CBR * (top + 2 * role + (i >> 2)) + (i & 1), CBW * sxi),
                                             (i & 1) ? ve ^ 64u : ve, 0u, coef_lds + 1024 * i);
            } else {                    // role 2: four pieces -- above / below (i >> 1), plane i & 1
#pragma unroll
                for (int i = 0; i < NHROW / 8; ++i)
                    lds_dma16_brun<1, false>((i & 1) ? srd_r : srd_b, chroma_off((i >> 1) ? CBR * (top + QS) : CBR * top - 1, CBW * sxi),
                                             (i & 1) ? ve ^ 64u : ve, 0u, coef_lds + 1024 * i);
            }
            return;
        }
        // role 3: the neighbour blocks left and right of the stack, block by block (two planes within one instruction:
        // per-lane addresses; a block outside the plane fetches block 0 and is not used)
#pragma unroll
        for (int i = 0; i < NDMA_C; ++i) {
            if (8 * i >= NSIDE) break;
            const int b = 8 * i + (lane >> 3);
            const int pl = (b >> 1) & 1, row = min(max(CBR * top - 1 + (b >> 2), 0), a.uyc - 1);
            const int bx = (b & 1) ? CBW * sxi + CBW : CBW * sxi - 1;
            const int16_t *cbase = a.coef[1 + pl] + img * a.coef_stride[1 + pl];
            const uint32_t blk = (bx >= 0 && bx < a.uxc) ? (uint32_t)row * a.uxc + bx : 0u;
            const int c = (lane & 7) ^ ((b >> 1) & 7);
            lds_dma16_keep(reinterpret_cast<const char *>(cbase) + ((size_t)blk * 128 + 16 * c), coef_lds + 1024 * i);
        }
    };

#ifdef JA_X_STAGGER   // experiment: the three workgroups of a CU start a third of a strip apart
    for (int d = (int)(blockIdx.x / 256u) * (JA_X_STAGGER); d > 0; d -= 64 * 100) __builtin_amdgcn_s_sleep(100);
#endif
    if ((int)blockIdx.x >= a.nstacks) return;   // (the grid is never larger than the call)
    // The walk.  Static (a.tickets == nullptr): workgroup b takes stacks b, b + grid, ...  Dynamic, for calls that are many
    // trips long: the first stack is b, every further one a ticket from a global counter -- wave 0 draws it at the top of a
    // trip and publishes it in LDS before its arrival at "ready", the others pick it up behind their wait for "ready".  A
    // SIMD issues its oldest ready wave first, so the workgroups a CU received first run faster than the later ones
    // (512 x 1080p, static: the three generations end at 77 / 87 / 100 % of the call, profiles/r03_ab_generation_priority.txt);
    // with tickets the fast ones simply walk more stacks and everybody leaves together (-5 %).  Short walks keep the
    // static order: their last round is better planned than drawn (8192 x 8192, 5.33 trips: +5 % with tickets).
    constexpr bool dyn = DYN;   // the walk is a template parameter: each variant carries only its own branch and scalar state
    int cur = (int)blockIdx.x;
    const int last_round = a.nstacks - (int)gridDim.x, second_last_round = a.nstacks - 2 * (int)gridDim.x;
    int c_img, c_syi, c_sxi;   // the current trip's stack (located once, a trip ahead)
    locate(cur, c_img, c_syi, c_sxi);
    dma_chroma(c_img, c_syi, c_sxi, lane0, role_of(0));
    int img_of_table = -1;
    int stores_behind_dma = 0;  // wave-uniform
    JA_PHASE_DECL

    for (int trip = 0;; ++trip) {
        // Launder the lane id once per strip: everything below that depends only on the lane is
        // cheap to recompute, but hoisted out of this loop it would pin ~60 VGPRs for good.
        int lane = lane0;
        asm volatile("" : "+v"(lane));
        const int lbx = lane & (BX - 1), seg = (int)((unsigned)lane / BX);
        const int img = c_img, syi = c_syi, sxi = c_sxi;
        // a strip below the image (the last stack of an image may be short): the wave still plays its role in the chroma
        // pass and keeps the counters, but has no luma blocks and no pixels
        const bool phantom = syi >= strips_y;

        // ---- modulated tables (only when they change: per image, or never when the batch shares one set) ----
        const int table_id = a.quanta_stride == 0 ? 0 : img;
        if (table_id != img_of_table) {
            const int qk = lane & 7, qh = lane >> 3;
#pragma unroll
            for (int p = 0; p < 3; ++p)
                sqw[qp][p][8 * qk + qh] = modulate_entry(qk, qh, 0.125f, a.quanta[img * a.quanta_stride + 64 * a.qi[p] + zigzag_of(qk, qh)]);
            img_of_table = table_id;
        }

        // ---- the chroma pass's coefficients: wait for the DMA, read 8 x 16 B (swizzled).  VM operations retire in issue
        //      order and the DMA was issued BEFORE the previous strip's pixel stores: when that strip took the branch-free
        //      store path (exactly 2 store instructions per pixel row) only the DMA has to be waited for ----
        if (stores_behind_dma == 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        JA_PHASE(0)
        // The scheduler of a SIMD issues its oldest ready wave first.  From the start of a strip to the arrival at the
        // "ready" counter a wave runs at the top priority (whoever arrives late is waited for by the others);
        // after it at most at priority 2, by strips left -- laggards catch up and the waves of a SIMD leave together.
        __builtin_amdgcn_s_setprio(3);
        uint32_t ticket = 0;
        if (dyn && qp == 0 && lane0 == 0) ticket = __hip_atomic_fetch_add(a.tickets, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ONE lane
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
        const uint32_t done_seen = lds_peek(done);   // checked after the transform; read here so that the check costs no round trip
        // w holds the chroma pass's block; the luma blocks of the strip follow it into the buffer
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (!phantom) dma_luma(img, syi, sxi, lane);
        __builtin_amdgcn_sched_barrier(0);
        JA_PHASE(1)

        const int top = syi - qp;
        const bool stack_above = top > 0, stack_below = CBR * (top + QS) < a.uyc;   // uniform over the stack
        const bool has_left = sxi > 0, has_right = CBW * sxi + CBW < a.uxc;
        const int first_bad = (a.uxc << 1) - sxi * (CW / 4) + 1;   // first tile dword past the plane (PITCH - 1 at a full last tile)
        // the plane's left / right edge: the reference clamps the sample index (decode.swift:4245)
        auto fix_columns = [&](uint32_t *row) {
            if (!has_left) row[0] = (row[1] & 0xffu) * 0x01010101u;
            if (first_bad < PITCH) {
                const uint32_t last = (row[first_bad - 1] >> 24) * 0x01010101u;
                for (int c = first_bad; c < PITCH; ++c) row[c] = last;
            }
        };
        {
            // ---- the chroma pass, by role.  Three complete code paths with nothing merged behind them: with a common
            //      tail LLVM keeps the values of several transforms alive across the branches and spills.  clamp [0, 255]
            //      + truncate == the saturating convert under round-toward-zero (trunc_pack*, fused_common.hpp).  Before its samples go into the tile a wave checks that
            //      everyone has read the previous trip's: the others signalled "done" two pixel rows before the end of
            //      their previous strip, more than a transform ago -- this rarely waits. ----
            const int role = role_of(trip);
            const int pl = role < 2 ? (lane >> 4) & 1 : role == 2 ? (lane / CBW) & 1 : (lane >> 1) & 1;
            uint32_t *tile = qt + pl * PLANE;
            if (role < 2) {
                float g[64];
#ifdef JA_X_NOCIDCT   // experiment (wrong pixels): the walk without the arithmetic of its chroma transform
#pragma unroll
                for (int i = 0; i < 64; ++i) g[i] = (float)(w[i & 31] >> (i & 32 ? 16 : 0) & 0xff) + sqw[qp][1 + pl][i];
#else
                idct_block(w, TransposedTable{sqw[qp][1 + pl]}, 128.5f, g);
#endif
                uint32_t pk[16];
                trunc_pack24(g, pk); trunc_pack24(g + 24, pk + 6); trunc_pack16(g + 48, pk + 12);
#pragma unroll
                for (int i = 0; i < 16; ++i) asm volatile("" : "+v"(pk[i]));
                JA_PHASE(2)
                lds_wait_ge_seen(done, (uint32_t)(QS * trip), done_seen);
                JA_PHASE(3)
                const int idx = lane & 15;
                uint32_t *dst = tile + (1 + CR * (2 * role + (lane >> 5)) + 8 * (idx / CBW)) * PITCH + 1 + 2 * (idx % CBW);
#pragma unroll
                for (int y = 0; y < 8; ++y) { dst[y * PITCH] = pk[2 * y]; dst[y * PITCH + 1] = pk[2 * y + 1]; }
                // the edge columns of the rows this wave has just written, at the plane's left / right edge
                if (!has_left || first_bad < PITCH) {
                    if (lane < 4 * CR) fix_columns(qt + (lane / (2 * CR)) * PLANE + (1 + 2 * CR * role + lane % (2 * CR)) * PITCH);
                }
            } else if (role == 2) {   // above: the block's last sample row; below: its first
                const bool below = lane >= 2 * CBW;
                float r[8];
                // above the stack: the block's LAST sample row (sign flip of idct8's s0), below it: the first -- from the lane's
                // place, arithmetically: (lane - 2 CBW) is negative exactly for the lanes above
                idct_block_edge_row(w, TransposedTable{sqw[qp][1 + pl]}, 128.5f, (uint32_t)(lane - 2 * CBW) & 0x80000000u, r);
                uint32_t p01[2];
                trunc_pack8(r, p01);
                const uint32_t p0 = p01[0], p1 = p01[1];
                JA_PHASE(2)
                lds_wait_ge_seen(done, (uint32_t)(QS * trip), done_seen);
                JA_PHASE(3)
                // (the lanes of the rows that exist, as a scalar range: no per-lane select of the two flags)
                const int lo = stack_above ? 0 : 2 * CBW, hi = stack_below ? NHROW : 2 * CBW;
                if (lane >= lo && lane < hi) {
                    uint32_t *dst = tile + (below ? QROWS - 1 : 0) * PITCH + 1 + 2 * (lane % CBW);
                    dst[0] = p0; dst[1] = p1;
                }
            } else {   // neighbour blocks: the column that touches the stack (first of the right, last of the left neighbour)
                const int rowi = lane >> 2, side = lane & 1;
                uint32_t e[8];   // the edge sample of each row, replicated: column 0 of the right neighbour (side 1), column 7 of the left one
                {
                    float edge[8];
                    idct_block_edge_col(w, TransposedTable{sqw[qp][1 + pl]}, 128.5f, side ? 0u : 0x80000000u, edge);
                    trunc_bytes8(edge, e);
#pragma unroll
                    for (int y = 0; y < 8; ++y) e[y] *= 0x01010101u;
                }
#pragma unroll
                for (int y = 0; y < 8; ++y) asm volatile("" : "+v"(e[y]));
                JA_PHASE(2)
                lds_wait_ge_seen(done, (uint32_t)(QS * trip), done_seen);
                JA_PHASE(3)
                const uint32_t sides = (has_left ? 1u : 0u) | (has_right ? 2u : 0u);   // scalar; bit `side` of it, per lane
                if (lane < NSIDE && ((sides >> side) & 1u)) {
                    uint32_t *col = tile + (side ? PITCH - 1 : 0);
                    if (rowi == 0) { if (stack_above) col[0] = e[7]; }                                  // corner samples
                    else if (rowi == QS * CBR + 1) { if (stack_below) col[(QROWS - 1) * PITCH] = e[0]; }
                    else {
#pragma unroll
                        for (int y = 0; y < 8; ++y) col[(1 + 8 * (rowi - 1) + y) * PITCH] = e[y];
                    }
                }
            }
        }
        // this wave's samples are in the tile: arrive, do not wait yet
        lds_arrive(ready);
        JA_PHASE(4)
        if (cur < second_last_round) __builtin_amdgcn_s_setprio(2);        // by rounds left after this stack
        else if (cur < last_round) __builtin_amdgcn_s_setprio(1);
        else __builtin_amdgcn_s_setprio(0);
        bool more = false;
        auto take = [&](int next) {   // the stack of the next trip: locate it and request the blocks of its chroma pass
            more = next < a.nstacks;
            if (more) {
                int n_img, n_syi, n_sxi;
                locate(next, n_img, n_syi, n_sxi); c_img = n_img; c_syi = n_syi; c_sxi = n_sxi;
                dma_chroma(n_img, n_syi, n_sxi, lane, role_of(trip + 1));
                cur = next;
            }
        };
        // dynamic walk: wave 0 publishes the drawn stack behind its wait for the luma blocks (the atomic was issued in front of
        // their DMA, and VM operations retire in order), the others wait for the publication -- which is a luma transform old
        // by the time they look
        auto drawn = [&]() -> int {
            lds_wait_ge(pub, (uint32_t)(trip + 1));
            return __builtin_amdgcn_readfirstlane((int)*(volatile uint32_t *)&next_slot[trip & 1]);
        };
        if (phantom) {   // nothing to decode: the next pass's blocks, both counters, next trip  (wave 0 is never phantom)
            if (dyn) take(drawn());
            else take(cur + (int)gridDim.x);
            lds_arrive(done);
            stores_behind_dma = 0;
            if (!more) break;
            continue;
        }
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        JA_PHASE(11)
        if (dyn && qp == 0) {
            if (lane0 == 0) *(volatile uint32_t *)&next_slot[trip & 1] = gridDim.x + ticket;
            lds_arrive(pub);
        }
        read_block();
        // ---- the coefficient buffer is consumed: prefetch the next stack's chroma pass into it.  From here to the end of
        //      the strip only stores are issued, so nothing waits on the DMA.  (One DMA instruction in front of each column of
        //      the transform instead of eight in a row, pinned with scheduling barriers, costs the transform more than the
        //      burst costs; one or two per pixel row change nothing: profiles/r03_ab_*dma*.txt.) ----
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        JA_PHASE(12)
        if (!dyn) take(cur + (int)gridDim.x);
        const uint32_t ready_seen = lds_peek(ready);
        __builtin_amdgcn_sched_barrier(0);
        JA_PHASE(13)

        // ---- luma: dequantise + IDCT, clamp + truncate (decode.swift:4121-4122), kept as integer-valued floats ----
        // the samples wait for their pixel row as packed bytes (16 registers instead of 64): clamp + truncate is the
        // saturating convert under round-toward-zero, and a row unpacks its eight bytes with v_cvt_f32_ubyte0..3
        uint32_t ypk[16];
        {
            float yv[64];
#ifdef JA_X_NOIDCT  // experiment (wrong pixels): how long is a strip without the IDCT arithmetic?
#pragma unroll
            for (int i = 0; i < 64; ++i) yv[i] = (float)(w[i & 31] >> (i & 32 ? 16 : 0) & 0xff);
#else
            idct_block(w, TransposedTable{sqw[qp][0]}, 128.5f, yv);
#endif
            trunc_pack24(yv, ypk); trunc_pack24(yv + 24, ypk + 6); trunc_pack16(yv + 48, ypk + 12);
        }
        // pin the IDCT here (LLVM otherwise sinks it into the pixel rows)
#pragma unroll
        for (int i = 0; i < 16; ++i) asm volatile("" : "+v"(ypk[i]));
        __builtin_amdgcn_sched_barrier(0);
        JA_PHASE(6)

        // ---- the stack's tile is complete: everyone's samples of this trip are in it (the wave arrived a luma transform
        //      ago; it was read then so that the check costs no round trip) ----
        lds_wait_ge_seen(ready, (uint32_t)(QS * (trip + 1)), ready_seen);
        if (dyn) take(drawn());   // the prefetch lands during the pixel rows (that costs nothing: profiles/r03_ab_ticket_walk.txt)
        JA_PHASE(9)
        {
            // the sample row above the wave's first / below its last where it is not in the plane: image top / bottom (a
            // missing row is the nearest own row, decode.swift:4246), and the edge columns of the stack's two halo rows
            // (decode.swift:4245).  Only this wave reads the rows it repairs.
            const int rows_avail = 8 * (a.uyc - CBR * syi);   // chroma sample rows of the plane from the first one under this strip
            auto copy_row = [&](int dst, int src) {          // rows of the wave's window, both planes
                for (int d = lane; d < 2 * PITCH; d += 64) {
                    uint32_t *col = sc + (d >= PITCH ? PLANE + d - PITCH : d);
                    col[dst * PITCH] = col[src * PITCH];
                }
            };
            if (syi == 0) copy_row(0, 1);
            else if (qp == 0 && (!has_left || first_bad < PITCH) && lane < 2) fix_columns(qt + lane * PLANE);
            if (rows_avail > 0 && rows_avail <= CR) copy_row(rows_avail + 1, rows_avail);
            else if (qp == QS - 1 && (!has_left || first_bad < PITCH) && lane < 2) fix_columns(qt + lane * PLANE + (QROWS - 1) * PITCH);
        }

        // ---- chroma rows, produced just in time from the tile.  Patch row j of a block: window row seg (8 / 2) + j;
        //      the LDS reads (hraw) and the conversion + horizontal interpolation (hconv) are separate so that the reads
        //      can be issued well ahead of their use ----
        auto hraw = [&](int pl, int j, uint32_t (&r)[3]) {
            const uint32_t *row = sc + pl * PLANE + (seg * 4 + j) * PITCH;
            r[0] = row[lbx]; r[1] = row[1 + lbx]; r[2] = row[2 + lbx];
        };
        auto hconv = [&](const uint32_t (&r)[3], float (&o)[8]) {   // samples enter as 2^15 + p + 1/32 (upsample.hpp)
            const float p[6] = {ubyte_magic<3>(r[0]), ubyte_magic<0>(r[1]), ubyte_magic<1>(r[1]),
                                ubyte_magic<2>(r[1]), ubyte_magic<3>(r[1]), ubyte_magic<0>(r[2])};
            lerp_row_2x(p, o);
        };
        // final chroma value of one pixel from the vertically combined sum: floor(v / 16 + 1/2) [- 128], without a floor
        auto finish = [&](float v) -> float {
            return __builtin_fmaf(v, 0.0625f, kMagic - 32768.0f - (MODE == 1 ? 128.0f : 0.0f)) - kMagic;
        };
        // sliding window over the patch rows: slots hold rows (y >> 1), (y >> 1) + 1, (y >> 1) + 2
        float hw[2][3][8];
        uint32_t rawn[2][3];
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) {
            uint32_t r0[3], r1[3];
            hraw(pl, 0, r0); hraw(pl, 1, r1); hraw(pl, 2, rawn[pl]);
            hconv(r0, hw[pl][0]); hconv(r1, hw[pl][1]);
        }

        // ---- store geometry: per pixel row the strip's BY segments are 96 chunks of 16 B; a lane stores chunk `lane` (and
        //      lanes 0..31 also chunk 64 + lane).  Both store instructions of a row cover whole 128-byte lines. ----
        const int tile_px = min(BX * 8, a.W - BX * 8 * sxi);    // pixels of this strip inside the image
        const int nb = 3 * tile_px;                              // bytes per row segment to write
        const uint32_t pitch = 3u * a.W;
        uint8_t *strip_out = a.out + img * a.out_stride + ((size_t)(8 * BY * syi) * a.W + BX * 8 * sxi) * 3;
        int sg0, sg1;   // segments of chunk `lane` and of chunk 64 + lane (the latter for lanes 0..31)
        if constexpr (BX == 32) { sg0 = 1 + ((lane - 48) >> 31); sg1 = 1; }   // (lane >= 48 ? 1 : 0, without a select)
        else { sg0 = (int)((unsigned)lane / CPS); sg1 = (int)((64u + (unsigned)lane) / CPS); }
        const int j0 = lane - CPS * sg0, j1 = 64 + lane - CPS * sg1;
        // inside the image: 16 j < nb (and, for the second chunk, lane < 32) -- as the SIGN of a difference, so that what depends on it
        // is formed with shifts and v_bfi instead of v_cndmask_b32 (ten times slower than either on gfx950)
        const int in0 = 16 * j0 - nb, in1 = max(16 * j1 - nb, lane - 32);            // negative: inside
        const bool col0 = in0 < 0, col1 = in1 < 0;
        // FAST: the strip's rows through a BUFFER RESOURCE (round 5).  base = the strip's first pixel, num_records = the bytes from
        // there to the end of the strip's last row INSIDE the image: a row below the image is out of range and the hardware drops
        // its store (range check: voffset >= num_records - soffset, tools/probe_buffer.hip); a chunk right of the image gets a
        // voffset that is out of range for every row.  No predicate, no branch and no 64-bit address per row: the row is the
        // scalar soffset.  (num_records <= 32 rows x 3 x 65 535 B: the resource describes a strip, not the image, which may
        // exceed 4 GiB.)
        const int rows_here = min(8 * BY, a.H - 8 * BY * syi);
        const uint32_t strip_bytes = (uint32_t)rows_here * pitch;
        const i32x4_t out_srd = make_srd(strip_out, strip_bytes);
        // FAST: every chunk is whole or outside.  Otherwise (any width: rows start at any byte address, which buffer stores
        // take) a chunk is whole, outside, or -- in the strip column that holds the image's right edge -- the PARTIAL last chunk
        // of its row segment (1 .. 15 bytes), which goes dword- and byte-wise (store_tail below).
        const int rem0 = col0 ? nb - 16 * j0 : 0, rem1 = col1 ? nb - 16 * j1 : 0;   // bytes of the lane's chunks inside the image
        const uint32_t base0 = sg0 * 8u * pitch + 16u * j0, base1 = sg1 * 8u * pitch + 16u * j1;
        // voffset = the chunk's place, or an out-of-range value: base where d < 0, 0x80000000 elsewhere (d: FAST in0 / in1; else "fewer than
        // 16 bytes of the chunk are inside")
        auto place = [](uint32_t base, int d) -> uint32_t {
            const uint32_t m = (uint32_t)(d >> 31);               // all ones where d < 0
            return (base & m) | (0x80000000u & ~m);               // one v_bfi_b32
        };
        const uint32_t voff0 = place(base0, FAST ? in0 : in0 + 15), voff1 = place(base1, FAST ? in1 : max(16 * j1 - nb + 15, lane - 32));
        stores_behind_dma = FAST ? 16 : 0;
        JA_PHASE(7)

        // One pixel row of the strip's BY block rows at a time, software-pipelined: row y is staged (ds_write) and read
        // back as 16-byte chunks (ds_read) right after its arithmetic, but the chunks are stored only after the arithmetic
        // of the next row; the chroma dwords of the next patch row are requested two rows ahead.
        uint4 pv0 = make_uint4(0, 0, 0, 0), pv1 = make_uint4(0, 0, 0, 0);   // chunks of the previous pixel row
        // FAST = false, right-edge column: the first `rem` (1 .. 15) bytes of a chunk -- up to three dwords, then up to three
        // bytes; a lane without a partial chunk carries out-of-range offsets throughout (rem outside 1 .. 15)
        auto store_tail = [&](const uint4 &v, uint32_t base, int rem, uint32_t soff) {
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
                // chunk `lane` from every lane, chunk 64 + lane from lanes 0 .. 31 (the others carry an out-of-range voffset):
                // exactly two store instructions per pixel row, whatever the strip's place in the image
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
#pragma unroll
        for (int y = 0; y < 8; ++y) {  // pixel row y of every block row of the strip
            __builtin_amdgcn_sched_barrier(0);
            // vertical interpolation: the nearer patch row (the middle of the window) weighs 3, the farther one (above
            // for even y, below for odd) 1; then colour and pack
            float cv[2][8];
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) {
                if ((y & 1) == 1) hconv(rawn[pl], hw[pl][2]);
#pragma unroll
                for (int x = 0; x < 8; ++x) cv[pl][x] = finish(w31(hw[pl][1][x], hw[pl][(y & 1) ? 2 : 0][x]));
                if ((y & 1) == 1) {
#pragma unroll
                    for (int x = 0; x < 8; ++x) { hw[pl][0][x] = hw[pl][1][x]; hw[pl][1][x] = hw[pl][2][x]; }
                }
            }
            uint32_t d[6];
            {
                float c[24];
#pragma unroll
                for (int x = 0; x < 8; ++x) {
                    const float yy = x == 0 ? ubyte<0>(ypk[2 * y]) : x == 1 ? ubyte<1>(ypk[2 * y]) : x == 2 ? ubyte<2>(ypk[2 * y]) : x == 3 ? ubyte<3>(ypk[2 * y])
                                   : x == 4 ? ubyte<0>(ypk[2 * y + 1]) : x == 5 ? ubyte<1>(ypk[2 * y + 1]) : x == 6 ? ubyte<2>(ypk[2 * y + 1]) : ubyte<3>(ypk[2 * y + 1]);
                    if constexpr (MODE == 1) {
                        const float pb = cv[0][x], pr = cv[1][x];
                        // jpeg.swift:441-453: x = (y + m_cb cb) + m_cr cr, clamped and TRUNCATED -- the pack below.  One FMA
                        // for R and B, two for G: after the truncation every one of the 2^16 / 2^24 input combinations gives
                        // the reference's byte (tests/test_colour_rounding.py enumerates them).
                        c[3 * x + 0] = __builtin_fmaf(1.40200f, pr, yy);
                        c[3 * x + 1] = __builtin_fmaf(-0.71414f, pr, __builtin_fmaf(-0.34414f, pb, yy));
                        c[3 * x + 2] = __builtin_fmaf(1.77200f, pb, yy);
                    } else {
                        c[3 * x + 0] = yy; c[3 * x + 1] = cv[0][x]; c[3 * x + 2] = cv[1][x];
                    }
                }
                trunc_pack24(c, d);   // clamp [0, 255] + truncate
            }
            __builtin_amdgcn_sched_barrier(0);
            // the row's traffic: the previous row's stores, this row's staging, the request for the next patch row
            if (y > 0) store_row(y - 1);
            uint2 *sw = reinterpret_cast<uint2 *>(stage_w + seg * SEG_DW + lbx * 6);
            sw[0] = make_uint2(d[0], d[1]);
            sw[1] = make_uint2(d[2], d[3]);
            sw[2] = make_uint2(d[4], d[5]);
            pv0 = *reinterpret_cast<const uint4 *>(stage_w + 4 * lane);
            pv1 = *reinterpret_cast<const uint4 *>(stage_w + 4 * (64 + (lane & 31)));
            if ((y & 1) == 1 && y < 7) {
#pragma unroll
                for (int pl = 0; pl < 2; ++pl) hraw(pl, (y >> 1) + 3, rawn[pl]);
                // after the request for patch row 5 the wave reads nothing more from the tile (the LDS performs the reads
                // before the add)
                if (y == 5) lds_arrive(done);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        store_row(7);
        JA_PHASE(10)
        if (!more) break;
    }
    JA_PHASE_FLUSH((int)blockIdx.x * NW + qp, lane0)
}

// Persistent grid = what is resident at once (LDS-bound: three workgroups per CU) -- of the instantiation that is launched:
// the two walks (DYN) of a shape are separate kernels with their own register counts.
template <int MODE, int BX, bool FAST, bool DYN>
int quad_resident_workgroups()
{
    return resident_workgroups_of<k_quad420<MODE, BX, FAST, DYN>>(2);
}

template <int MODE, int BX, bool FAST, bool DYN>
hipError_t launch_quad_walk(hipStream_t stream, const QuadArgs &a)
{
    int cap = quad_resident_workgroups<MODE, BX, FAST, DYN>();
#ifdef JA_X_GRID_PER_CU   // experiment: fewer resident workgroups per CU than fit
    cap = std::min(cap, (JA_X_GRID_PER_CU) * (cap / 3));
#endif
#ifdef JA_PHASE_PROFILE   // development aid: JA_GRID_CAP=256 runs one workgroup per CU (a wave alone on its SIMD)
    if (const char *e = std::getenv("JA_GRID_CAP")) cap = std::min(cap, std::atoi(e));
#endif
    hipLaunchKernelGGL((k_quad420<MODE, BX, FAST, DYN>), dim3(std::min(a.nstacks, cap)), dim3(kThreads), 0, stream, a);
    return hipGetLastError();
}
template <int MODE, int BX, bool FAST>
hipError_t launch_quad(hipStream_t stream, const QuadArgs &a)
{
    return a.tickets ? launch_quad_walk<MODE, BX, FAST, true>(stream, a) : launch_quad_walk<MODE, BX, FAST, false>(stream, a);
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

bool quad_decode_supported(const jpeg_amd_layout &L)
{
    if (L.nplanes != 3 || L.precision != 8 || L.scale_x != 2 || L.scale_y != 2) return false;
    if (L.factor_x[0] != 2 || L.factor_y[0] != 2) return false;
    for (int p = 1; p < 3; ++p)
        if (L.factor_x[p] != 1 || L.factor_y[p] != 1) return false;
    return L.units_x[1] == L.units_x[2] && L.units_y[1] == L.units_y[2];
}

// How a plane is cut into strip columns.  A stack costs a workgroup the same whether its strips are full, half empty
// (a partial last column) or phantom (a short last stack), so the cut that needs the fewest stacks wins: 32 x 2 strips
// throughout, 16 x 4 strips throughout, or -- for batches, where a second launch is free -- 32 x 2 strips for the whole
// columns and ONE column of 16 x 4 strips for a remainder of at most 16 blocks (1920 x 1080: 7 columns x 17 stacks + 1 x 9
// = 128 stacks per image for 126.6 stacks' worth of blocks, where 15 columns of 16 x 4 strips need 135 and 8 of 32 x 2 136).
struct QuadCut { int parts; int bx[2], sx0[2], cols[2]; };
static long quad_stacks(int cols, int uy, int by) { return (long)cols * (((uy + by - 1) / by + 3) / 4); }
QuadCut quad_cut(int ux, int uy, long n_images, long resident)
{
#ifdef JA_X_FORCE_BX
    return QuadCut{1, {JA_X_FORCE_BX, 0}, {0, 0}, {(ux + JA_X_FORCE_BX - 1) / JA_X_FORCE_BX, 0}};
#endif
    const long wide = quad_stacks((ux + 31) / 32, uy, 2), narrow = quad_stacks((ux + 15) / 16, uy, 4);
    QuadCut best = narrow < wide ? QuadCut{1, {16, 0}, {0, 0}, {(ux + 15) / 16, 0}} : QuadCut{1, {32, 0}, {0, 0}, {(ux + 31) / 32, 0}};
    const int whole = ux / 32, rest = ux - 32 * whole;
    if (whole > 0 && rest > 0 && rest <= 16) {
        const long mixed = quad_stacks(whole, uy, 2) + quad_stacks(1, uy, 4);
        // worth a second launch only when the call is many trips long (the narrow column's launch is at least one trip)
        if (mixed < std::min(wide, narrow) && n_images * std::min(wide, narrow) >= 16 * resident)
            best = QuadCut{2, {32, 16}, {0, 2 * whole}, {whole, 1}};
    }
    return best;
}

hipError_t launch_quad_decode(hipStream_t stream, int n_images, const jpeg_amd_layout &L, const PlaneSet &coef, QuantaRef q,
                              bool rgb, uint32_t *d_walk_counters, uint8_t *d_pixels, size_t pixel_stride)
{
    QuadArgs a{};
    for (int p = 0; p < 3; ++p) {
        a.coef[p] = static_cast<const int16_t *>(coef.ptr[p]);
        a.coef_stride[p] = coef.stride[p];
        a.qi[p] = L.qi[p];
    }
    a.quanta = q.d_quanta; a.quanta_stride = q.image_stride;
    a.ux = L.units_x[0]; a.uy = L.units_y[0];
    a.uxc = L.units_x[1]; a.uyc = L.units_y[1];
    a.W = L.width; a.H = L.height;
    a.out = d_pixels; a.out_stride = pixel_stride;
    if (n_images == 0 || a.ux == 0 || a.uy == 0) return hipSuccess;
    const bool fast = (L.width & 15) == 0 && (pixel_stride & 15) == 0 && (reinterpret_cast<uintptr_t>(d_pixels) & 15) == 0;
    // every decision is sized from the instantiation it concerns (colour target x strip shape x store path)
    auto resident = [&](int bx) -> int {
        // (the ticket walk's residency: it is the walk whose length the threshold below and the mixed cut are about)
        if (bx == 16) return fast ? (rgb ? quad_resident_workgroups<1, 16, true, true>() : quad_resident_workgroups<0, 16, true, true>())
                                  : (rgb ? quad_resident_workgroups<1, 16, false, true>() : quad_resident_workgroups<0, 16, false, true>());
        return fast ? (rgb ? quad_resident_workgroups<1, 32, true, true>() : quad_resident_workgroups<0, 32, true, true>())
                    : (rgb ? quad_resident_workgroups<1, 32, false, true>() : quad_resident_workgroups<0, 32, false, true>());
    };
    const QuadCut cut = quad_cut(a.ux, a.uy, n_images, std::max(resident(32), resident(16)));
    for (int part = 0; part < cut.parts; ++part) {
        const int bx = cut.bx[part];
        a.tiles_x = cut.cols[part]; a.sx0 = cut.sx0[part];
        a.stacks_per_image = (int)quad_stacks(a.tiles_x, a.uy, 64 / bx);
        const long nstacks = (long)a.stacks_per_image * n_images;
        if (nstacks > 0x3fffffffL) return hipErrorInvalidValue;
        a.nstacks = (int)nstacks;
        // tickets for walks of at least 16 trips (kernel header: shorter ones are better planned than drawn)
#ifndef JA_X_TICKET_TRIPS
#define JA_X_TICKET_TRIPS 16
#endif
        a.tickets = nstacks >= (long)(JA_X_TICKET_TRIPS) * resident(bx) ? d_walk_counters : nullptr;
        if (a.tickets) {   // (a 2 us node in front of a call of a millisecond or more)
            const hipError_t m = hipMemsetAsync(a.tickets, 0, sizeof(uint32_t), stream);
            if (m != hipSuccess) return m;
        }
        hipError_t e;
        if (bx == 16) e = fast ? (rgb ? launch_quad<1, 16, true>(stream, a) : launch_quad<0, 16, true>(stream, a))
                               : (rgb ? launch_quad<1, 16, false>(stream, a) : launch_quad<0, 16, false>(stream, a));
        else e = fast ? (rgb ? launch_quad<1, 32, true>(stream, a) : launch_quad<0, 32, true>(stream, a))
                      : (rgb ? launch_quad<1, 32, false>(stream, a) : launch_quad<0, 32, false>(stream, a));
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

}  // namespace jpeg_amd
