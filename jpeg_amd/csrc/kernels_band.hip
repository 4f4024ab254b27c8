// kernels_band.hip -- single-launch fused Spectral -> pixels for ycc8 4:2:0 (and 4:4:0) without a
// chroma round trip through HBM: "band walk".
//
// Replaces idct() -> interleaved(cosite: false) -> unpack(as:) (decode.swift:4154, 4182, 4294) like
// kernels_fused.hip, for images / batches that are large enough to give every resident wave its
// own piece.  Differences to k_chroma_idct + k_luma_fused:
//
//   * every chroma block is transformed ONCE, inside the kernel that also consumes it; its samples
//     only ever live in LDS (no uint8 chroma planes in HBM: 403 MB instead of 485 MB of traffic for
//     one 8192 x 8192 image);
//   * a wave owns a BAND of 64 luma blocks (512 px; 32 chroma blocks per plane) and walks DOWN it.
//     One step = one chroma block row: a chroma pass (64 work-items = 2 planes x 32 blocks, all
//     busy) and two luma passes (64 blocks of one luma block row each) -- 1.5 IDCT passes per 64
//     luma blocks, the minimum for 4:2:0;
//   * the bilinear filter reaches one chroma sample up / down (decode.swift:4243-4257).  Walking
//     down, the row above is simply still there: the wave keeps the last 16 chroma sample rows in
//     an LDS ring.  The row below is produced by running the chroma pass one step AHEAD of the
//     second luma pass of a step;
//   * left / right neighbours of the band (one sample column each) come from a HALO pass at the
//     start of a piece: the (R + 2) x 2 sides x 2 planes neighbour blocks of the piece's R chroma
//     block rows are transformed together (<= 64 work-items) and their edge columns parked in LDS;
//   * a piece (unit of work) is R chroma block rows of one band; its first and last row need the
//     sample row above / below, which belongs to another piece: one EDGE pass each (the whole block
//     row is transformed and one sample row kept -- a separate one-row transform made LLVM merge
//     the two instruction streams and spill).
//
// Per piece: HALO + EDGE(top) + R x (chroma + 2 luma) + EDGE(bottom) passes.  R is chosen by the
// host so that the pieces spread evenly over the resident waves (an 8192 x 8192 image on 3 072
// waves: R = 3, 2 736 pieces, one per wave -- no tail of half-idle rounds).
//
// Everything a wave touches in LDS is private to it (ring 8.5 KiB, staging row 1.5 KiB, tables,
// halo columns): no workgroup barrier.  Coefficients are loaded by the work-item that transforms
// them (8 x 16 B of its own 128-byte line); with three waves per SIMD the load latency of one wave
// is covered by the arithmetic of the other two.
//
// Arithmetic: dct.hpp / upsample.hpp, the same operations in the same order as kernels_fused.hip
// (the exactness arguments are in that file's header and in tests/test_colour_rounding.py).
#pragma clang fp contract(off)

#include "dct.hpp"
#include "kernels.hpp"
#include "upsample.hpp"

#include <cstdlib>

namespace jpeg_amd {

namespace {

constexpr int kThreads = 256;
constexpr int NW = kThreads / 64;
constexpr int kBandBlocks = 64;          // luma blocks per band row
constexpr int kMaxRows = 14;             // (R + 2) * 4 halo blocks must fit one wave
constexpr int PITCH = 68;                // ring row: [0] pad, [1] left halo, [2..65] 256 samples, [66] right halo, [67] pad
constexpr int RING = 16;                 // sample rows per plane

struct BandArgs {
    const int16_t *coef[3];
    size_t coef_stride[3];
    const uint16_t *quanta;
    size_t quanta_stride;
    int qi[3];
    int ux, uy;        // luma units
    int uxc, uyc;      // chroma units
    int W, H;
    uint8_t *out;
    size_t out_stride;
    int nbands;        // bands per image
    int nsegs;         // pieces per band
    int R;             // chroma block rows per piece
    int total_units;
};

enum : int { K_HALO = 0, K_EDGE = 1, K_CHROMA = 2, K_LUMA = 3 };

__device__ __forceinline__ uint32_t pack4(const float *v)
{
    // clamp [0, 255] + truncate == saturating convert of floor(v) (decode.swift:4121-4122)
    uint32_t d = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) d = __builtin_amdgcn_cvt_pk_u8_f32(floorf(v[i]), i, d);
    return d;
}

// SX: horizontal chroma subsampling (2: 4:2:0, 1: 4:4:0 -- not yet instantiated); MODE 0 YCbCr bytes, 1 RGB bytes;
// FAST: W % 16 == 0 and 16-byte aligned rows (whole 16-byte chunks are inside or outside the image).
template <int MODE, bool FAST>
__global__ __launch_bounds__(kThreads, 3) void k_band420(BandArgs a)
{
    __shared__ __attribute__((aligned(16))) uint32_t ringw[NW][2][RING][PITCH];
    __shared__ __attribute__((aligned(16))) uint32_t stagew[NW][kBandBlocks * 6];
    __shared__ __attribute__((aligned(16))) float sqw[NW][3][64];
    __shared__ __attribute__((aligned(8))) uint32_t halow[NW][64 * 2];

    const int lane0 = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    uint32_t (*ring)[RING][PITCH] = ringw[wave];
    uint32_t *stage_w = stagew[wave];
    uint32_t *halo = halow[wave];
    const uint8_t *halo8 = reinterpret_cast<const uint8_t *>(halo);

    const int nwaves = gridDim.x * NW;
    const int units_per_image = a.nbands * a.nsegs;
    int img_of_table = -1;

    for (int u = blockIdx.x * NW + wave; u < a.total_units; u += nwaves) {
        int lane = lane0;
        asm volatile("" : "+v"(lane));   // keep lane-derived values out of long-lived registers
        const int img = u / units_per_image;
        const int rem = u - img * units_per_image;
        const int seg = rem / a.nbands, band = rem - seg * a.nbands;
        const int r0 = seg * a.R, r1 = min(r0 + a.R, a.uyc);
        const int nrows = r1 - r0;
        const bool has_left = band > 0, has_right = 32 * band + 32 < a.uxc;
        const int nvalid_c = min(32, a.uxc - 32 * band);          // chroma blocks of this band inside the plane
        const int fill_from = has_right ? PITCH : 2 + 2 * nvalid_c;   // first ring dword past the plane (decode.swift:4245: index clamp)

        if (img != img_of_table) {   // modulated tables (decode.swift:3984-4017), natural order
            const int qk = lane & 7, qh = lane >> 3;
#pragma unroll
            for (int t = 0; t < 3; ++t)
                sqw[wave][t][lane] = modulate_entry(qk, qh, 0.125f,
                                                    a.quanta[img * a.quanta_stride + 64 * a.qi[t] + zigzag_of(qk, qh)]);
            img_of_table = img;
        }

        const int nsteps = 3 + 3 * nrows;
        // which pass is step st (all wave-uniform); false: the step does not exist for this piece
        auto decode = [&](int st, int &kind, int &row, int &half, bool &last) -> bool {
            half = 0; last = false;
            if (st == 0) { kind = K_HALO; row = 0; return has_left || has_right; }
            if (st == 1) { kind = K_EDGE; row = r0 - 1; last = true; return r0 > 0; }
            if (st == 2) { kind = K_CHROMA; row = r0; return true; }
            const int i = (st - 3) / 3, ph = (st - 3) - 3 * i, cr = r0 + i;
            if (ph == 0) { kind = K_LUMA; row = 2 * cr; return true; }
            if (ph == 1) {
                if (cr + 1 < r1) { kind = K_CHROMA; row = cr + 1; return true; }
                kind = K_EDGE; row = cr + 1;
                return cr + 1 < a.uyc;
            }
            kind = K_LUMA; row = 2 * cr + 1; half = 1;
            return row < a.uy;
        };
        // the 128 bytes of coefficients of the block this work-item transforms in a pass
        auto block_of = [&](int kind, int row, int ln) -> const uint4 * {
            int pl = 0;
            size_t blk;
            if (kind == K_LUMA) {
                blk = (size_t)row * a.ux + min(kBandBlocks * band + ln, a.ux - 1);
            } else if (kind == K_HALO) {
                // work-item = (row - (r0 - 1)) * 4 + side * 2 + plane; blocks that do not exist read block 0
                const int rr = r0 - 1 + (ln >> 2);
                const int bx = ((ln >> 1) & 1) ? 32 * band + 32 : 32 * band - 1;
                pl = 1 + (ln & 1);
                const bool ok = (ln >> 2) < nrows + 2 && rr >= 0 && rr < a.uyc && bx >= 0 && bx < a.uxc;
                blk = ok ? (size_t)rr * a.uxc + bx : 0;
            } else {
                pl = 1 + (ln >> 5);
                blk = (size_t)row * a.uxc + min(32 * band + (ln & 31), a.uxc - 1);
            }
            const int16_t *base = (pl == 0 ? a.coef[0] : pl == 1 ? a.coef[1] : a.coef[2]) +
                                  img * (pl == 0 ? a.coef_stride[0] : pl == 1 ? a.coef_stride[1] : a.coef_stride[2]);
            return reinterpret_cast<const uint4 *>(base + 64 * blk);
        };
        uint32_t wn[32];   // the NEXT pass's block, requested while the current pass is being worked on
        auto fetch = [&](const uint4 *src) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const uint4 v = src[i];
                wn[4 * i + 0] = v.x; wn[4 * i + 1] = v.y; wn[4 * i + 2] = v.z; wn[4 * i + 3] = v.w;
            }
        };

        int st = 0, kind, row, half;
        bool last;
        while (!decode(st, kind, row, half, last)) ++st;   // step 2 always exists
        fetch(block_of(kind, row, lane));
#pragma unroll 1
        while (st < nsteps) {
            asm volatile("" : "+v"(lane));   // nothing derived from the lane id may be hoisted out of the step
            int st2 = st + 1, kind2 = 0, row2 = 0, half2 = 0;
            bool last2 = false;
            while (st2 < nsteps && !decode(st2, kind2, row2, half2, last2)) ++st2;
            const bool have_next = st2 < nsteps;

            uint32_t w[32];
#pragma unroll
            for (int i = 0; i < 32; ++i) w[i] = wn[i];
            const int pl = kind == K_LUMA ? 0 : kind == K_HALO ? 1 + (lane & 1) : 1 + (lane >> 5);
            const float *sq = sqw[wave][pl];

            float g[64];
            idct_block(w, sq, 128.5f, g);   // level = 2^(P-1) + 0.5, P = 8
            asm volatile("" : "+v"(lane));   // what the epilogues derive from the lane id must not be computed before the transform

            if (kind != K_LUMA) {
                if (have_next) fetch(block_of(kind2, row2, lane));
            }
            if (kind == K_EDGE) {
                // one sample row of the chroma block row above (its last row) / below (its first row) the piece
                const int pe = lane >> 5;
                float r[8];
#pragma unroll
                for (int x = 0; x < 8; ++x) r[x] = last ? g[56 + x] : g[x];
                const int rrow = (last ? 8 * row + 7 : 8 * row) & 15;
                uint32_t *dst = &ring[pe][rrow][2 + 2 * (lane & 31)];
                *reinterpret_cast<uint2 *>(dst) = make_uint2(pack4(&r[0]), pack4(&r[4]));
                // halo dwords and plane edge of that row
                if (lane < 2) {
                    uint32_t *rw = ring[lane][rrow];
                    const int hy = last ? 7 : 0;
                    const int hb = ((row - (r0 - 1)) * 4 + lane) * 8 + hy;
                    rw[1] = (has_left ? (uint32_t)halo8[hb] : (rw[2] & 0xffu)) * 0x01010101u;
                    if (has_right) rw[66] = (uint32_t)halo8[hb + 16] * 0x01010101u;
                    else {
                        const uint32_t lastv = (rw[fill_from - 1] >> 24) * 0x01010101u;
                        for (int c = fill_from; c < PITCH - 1; ++c) rw[c] = lastv;
                    }
                }
            } else if (kind == K_HALO) {
                // keep the neighbour block's edge column: its first column for the right neighbour, its last for the left one
                const int hside = (lane >> 1) & 1, rr = r0 - 1 + (lane >> 2);
                const int bx = hside ? 32 * band + 32 : 32 * band - 1;
                const bool active = (lane >> 2) < nrows + 2 && rr >= 0 && rr < a.uyc && bx >= 0 && bx < a.uxc;
                float col[8];
#pragma unroll
                for (int y = 0; y < 8; ++y) col[y] = hside ? g[8 * y] : g[8 * y + 7];
                if (active) *reinterpret_cast<uint2 *>(halo + 2 * lane) = make_uint2(pack4(&col[0]), pack4(&col[4]));
            } else if (kind == K_CHROMA) {
                const int rbase = (8 * row) & 15;
                uint32_t *dst = &ring[pl - 1][rbase][2 + 2 * (lane & 31)];
#pragma unroll
                for (int y = 0; y < 8; ++y)
                    *reinterpret_cast<uint2 *>(dst + y * PITCH) = make_uint2(pack4(&g[8 * y]), pack4(&g[8 * y + 4]));
                // halo dwords (neighbour columns, or the plane's own edge sample) and the plane's right edge
                if (lane < 16) {
                    const int hp = lane >> 3, y = lane & 7;
                    uint32_t *rw = ring[hp][rbase + y];
                    const int hb = ((row - (r0 - 1)) * 4 + hp) * 8 + y;
                    rw[1] = (has_left ? (uint32_t)halo8[hb] : (rw[2] & 0xffu)) * 0x01010101u;
                    if (has_right) rw[66] = (uint32_t)halo8[hb + 16] * 0x01010101u;
                    else {
                        const uint32_t lastv = (rw[fill_from - 1] >> 24) * 0x01010101u;
                        for (int c = fill_from; c < PITCH - 1; ++c) rw[c] = lastv;
                    }
                }
                if (row == 0) {   // top of the plane: the sample row above row 0 is row 0 (decode.swift:4240: t = 0)
                    for (int d = lane; d < 2 * PITCH; d += 64) {
                        const int p2 = d >= PITCH ? 1 : 0, c = d - p2 * PITCH;
                        ring[p2][15][c] = ring[p2][0][c];
                    }
                }
            } else {
            // ---- K_LUMA: clamp + truncate (decode.swift:4121-4122), kept as integer-valued floats ----
            float yv[64];
#pragma unroll
            for (int i = 0; i < 64; ++i) yv[i] = floorf(__builtin_amdgcn_fmed3f(g[i], 0.0f, 255.0f));
#pragma unroll
            for (int i = 0; i < 64; ++i) asm volatile("" : "+v"(yv[i]));
            __builtin_amdgcn_sched_barrier(0);

            const int cr = row >> 1;
            const int cbase = 8 * cr - 1 + 4 * half;   // ring row of patch row 0 (may be -1: & 15 wraps)
            if (half == 1 && cr + 1 >= a.uyc) {
                // bottom of the plane: the reference clamps the sample row index (decode.swift:4246)
                for (int d = lane; d < 2 * PITCH; d += 64) {
                    const int p2 = d >= PITCH ? 1 : 0, c = d - p2 * PITCH;
                    ring[p2][(8 * cr + 8) & 15][c] = ring[p2][(8 * cr + 7) & 15][c];
                }
            }
            constexpr float inv = 1.0f / 16.0f;
            constexpr float bias = MODE == 1 ? -127.5f : 0.5f;
            // horizontally interpolated chroma row j of this block's patch (x4)
            auto hrow = [&](int p2, int j, float (&o)[8]) {
                const uint32_t *rw = ring[p2][(cbase + j) & 15];
                const uint32_t d0 = rw[lane + 1], d1 = rw[lane + 2], d2 = rw[lane + 3];
                const float p[6] = {ubyte<3>(d0), ubyte<0>(d1), ubyte<1>(d1), ubyte<2>(d1), ubyte<3>(d1), ubyte<0>(d2)};
                lerp_row_2x(p, o);
            };
            auto finish = [&](float v) -> float { return floorf(__builtin_fmaf(v, inv, bias)); };

            float hw[2][3][8];   // patch rows j-1, j, j+1 of both planes (sliding window)
#pragma unroll
            for (int p2 = 0; p2 < 2; ++p2) { hrow(p2, 0, hw[p2][0]); hrow(p2, 1, hw[p2][1]); }

            // store geometry: one pixel row of the band is 96 chunks of 16 B; a lane stores chunk `lane`
            // and lanes 0..31 also chunk 64 + lane
            const int tile_px = min(8 * kBandBlocks, a.W - 8 * kBandBlocks * band);
            const int nb = 3 * tile_px;
            const uint32_t pitch = 3u * a.W;
            uint8_t *strip_out = a.out + img * a.out_stride + ((size_t)(8 * row) * a.W + 8 * kBandBlocks * band) * 3;
            const bool full = 8 * row + 8 <= a.H && tile_px == 8 * kBandBlocks;   // wave-uniform
            const bool col0 = 16 * lane < nb, col1 = lane < 32 && 16 * (64 + lane) < nb;

#pragma unroll
            for (int y = 0; y < 8; ++y) {
                if ((y & 1) == 0) __builtin_amdgcn_sched_barrier(0);
                if (y == 4) {   // half of the luma samples are consumed: their registers take the next pass's block
                    if (have_next) fetch(block_of(kind2, row2, lane));
                    __builtin_amdgcn_sched_barrier(0);
                }
                float cv[2][8];
#pragma unroll
                for (int p2 = 0; p2 < 2; ++p2) {
                    // window holds patch rows (y>>1), (y>>1)+1, (y>>1)+2; the nearer row (middle) weighs 3,
                    // the farther one (above for even y, below for odd) 1
                    if ((y & 1) == 1) hrow(p2, (y >> 1) + 2, hw[p2][2]);
#pragma unroll
                    for (int x = 0; x < 8; ++x) cv[p2][x] = finish(w31(hw[p2][1][x], hw[p2][(y & 1) ? 2 : 0][x]));
                    if ((y & 1) == 1) {
#pragma unroll
                        for (int x = 0; x < 8; ++x) { hw[p2][0][x] = hw[p2][1][x]; hw[p2][1][x] = hw[p2][2][x]; }
                    }
                }
                uint32_t d[6] = {0, 0, 0, 0, 0, 0};
#pragma unroll
                for (int x = 0; x < 8; ++x) {
                    const float yy = yv[8 * y + x];
                    float c0, c1, c2;
                    if constexpr (MODE == 1) {
                        // jpeg.swift:441-453; exactness of the fused forms: tests/test_colour_rounding.py
                        const float pb = cv[0][x], pr = cv[1][x];
                        const float yb = yy + kTruncBias;
                        c0 = __builtin_fmaf(1.40200f, pr, yb);
                        c1 = floorf(__builtin_fmaf(-0.71414f, pr, __builtin_fmaf(-0.34414f, pb, yy)));
                        c2 = __builtin_fmaf(1.77200f, pb, yb);
                    } else {
                        c0 = yy; c1 = cv[0][x]; c2 = cv[1][x];
                    }
                    d[(3 * x + 0) >> 2] = __builtin_amdgcn_cvt_pk_u8_f32(c0, (3 * x + 0) & 3, d[(3 * x + 0) >> 2]);
                    d[(3 * x + 1) >> 2] = __builtin_amdgcn_cvt_pk_u8_f32(c1, (3 * x + 1) & 3, d[(3 * x + 1) >> 2]);
                    d[(3 * x + 2) >> 2] = __builtin_amdgcn_cvt_pk_u8_f32(c2, (3 * x + 2) & 3, d[(3 * x + 2) >> 2]);
                }
                // stage the row (LDS ops of one wave execute in order), then store whole 16-byte chunks
                uint2 *sw = reinterpret_cast<uint2 *>(stage_w + lane * 6);
                sw[0] = make_uint2(d[0], d[1]);
                sw[1] = make_uint2(d[2], d[3]);
                sw[2] = make_uint2(d[4], d[5]);
                uint8_t *rowp = strip_out + (size_t)y * pitch;   // scalar
                const uint4 v0 = *reinterpret_cast<const uint4 *>(stage_w + 4 * lane);
                const uint4 v1 = *reinterpret_cast<const uint4 *>(stage_w + 4 * (64 + (lane & 31)));
                auto put = [&](uint8_t *o, const uint4 &v, int j) {
                    if constexpr (FAST) {
                        store_nt16(o, v);
                    } else {
                        const uint32_t vv[4] = {v.x, v.y, v.z, v.w};
                        for (int k = 0; k < 16; ++k)
                            if (16 * j + k < nb) o[k] = (uint8_t)(vv[k >> 2] >> (8 * (k & 3)));
                    }
                };
                if (FAST && full) {
                    put(rowp + 16 * lane, v0, lane);
                    if (lane < 32) put(rowp + 16 * (64 + lane), v1, 64 + lane);
                } else {
                    const bool rowok = 8 * row + y < a.H;
                    if (col0 && rowok) put(rowp + 16 * lane, v0, lane);
                    if (col1 && rowok) put(rowp + 16 * (64 + lane), v1, 64 + lane);
                }
            }
            }   // K_LUMA
            st = st2; kind = kind2; row = row2; half = half2; last = last2;
        }
    }
}

template <int MODE, bool FAST>
int band_resident_workgroups()
{
    static int cached = 0;
    if (cached == 0) {
        auto kernel = k_band420<MODE, FAST>;
        int per_cu = 0, dev = 0, cus = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, kThreads, 0) != hipSuccess || per_cu < 1) per_cu = 2;
        if (hipGetDevice(&dev) != hipSuccess ||
            hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1) cus = 256;
        cached = per_cu * cus;
    }
    return cached;
}

// 0 auto, 1 always (when the layout is supported), -1 never.  JPEG_AMD_BAND=1 / 0 is a development switch.
int band_override()
{
    static int v = [] {
        const char *e = std::getenv("JPEG_AMD_BAND");
        if (!e || !*e) return 0;
        return e[0] == '0' ? -1 : 1;
    }();
    return v;
}

}  // namespace

bool band_decode_supported(const jpeg_amd_layout &L, int n_images)
{
    if (band_override() < 0) return false;
    if (L.nplanes != 3 || L.scale_x != 2 || L.scale_y != 2) return false;
    if (n_images < 1 || L.units_x[1] < 1 || L.units_y[1] < 1) return false;
    // Measured (tools/ab_band.py): bit-identical, but 10-20 % slower than k_chroma_idct + k_luma_fused
    // at every size tried -- the step is bound by VALU issue, not by the 82 MB of chroma round trip this kernel
    // removes, and its halo / edge passes add 17 % more VALU instructions.  Kept as an opt-in.
    return band_override() > 0;
}

hipError_t launch_band_decode(hipStream_t stream, int n_images, const jpeg_amd_layout &L, const PlaneSet &coef,
                              QuantaRef q, bool rgb, uint8_t *d_pixels, size_t pixel_stride)
{
    BandArgs a{};
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
    a.nbands = (a.ux + kBandBlocks - 1) / kBandBlocks;
    const bool fast = (L.width & 15) == 0 && (pixel_stride & 15) == 0 && (reinterpret_cast<uintptr_t>(d_pixels) & 15) == 0;
    const int cap = rgb ? (fast ? band_resident_workgroups<1, true>() : band_resident_workgroups<1, false>())
                        : (fast ? band_resident_workgroups<0, true>() : band_resident_workgroups<0, false>());
    const long nwaves = (long)cap * NW;
    // rows per piece: the pieces should spread evenly over the resident waves.  cost ~ rounds x passes per piece
    // (3 per chroma block row + halo / edge passes)
    int best_r = 1;
    double best_cost = 1e300;
    for (int r = 1; r <= kMaxRows; ++r) {
        const long units = (long)n_images * a.nbands * ((a.uyc + r - 1) / r);
        const long rounds = (units + nwaves - 1) / nwaves;
        const double cost = (double)rounds * (3.0 * r + 1.4);
        if (cost <= best_cost) { best_cost = cost; best_r = r; }
    }
    a.R = best_r;
    a.nsegs = (a.uyc + a.R - 1) / a.R;
    const long total = (long)n_images * a.nbands * a.nsegs;
    if (total == 0) return hipSuccess;
    if (total > 0x7fffffffL) return hipErrorInvalidValue;
    a.total_units = (int)total;
    const long wgs_needed = (total + NW - 1) / NW;
    const int wgs = (int)(wgs_needed < cap ? wgs_needed : cap);
    if (rgb) {
        if (fast) hipLaunchKernelGGL((k_band420<1, true>), dim3(wgs), dim3(kThreads), 0, stream, a);
        else hipLaunchKernelGGL((k_band420<1, false>), dim3(wgs), dim3(kThreads), 0, stream, a);
    } else {
        if (fast) hipLaunchKernelGGL((k_band420<0, true>), dim3(wgs), dim3(kThreads), 0, stream, a);
        else hipLaunchKernelGGL((k_band420<0, false>), dim3(wgs), dim3(kThreads), 0, stream, a);
    }
    return hipGetLastError();
}

}  // namespace jpeg_amd
