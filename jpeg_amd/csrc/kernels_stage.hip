// kernels_stage.hip -- one gfx950 kernel per stage of the reference's spectral pipeline.
//
// These are the general kernels (any precision, any sampling factors, centred or
// cosited): they back the staged API (Spectral.idct / Planar.interleaved /
// Rectangular.unpack / pack / decomposed / Planar.fdct) one to one and are the fallback
// of the fused fast paths in kernels_fused.hip.
//
// Work decomposition for the DCT kernels: ONE 8x8 BLOCK PER WORK-ITEM.  A work-item keeps
// its 64 values in VGPRs, so both 1-D passes and both transposes of the reference
// (decode.swift:3971-3981, 4095-4099) are register renames -- no cross-lane traffic.
// Consecutive lanes own consecutive blocks of a block row, so row r of 64 neighbouring
// blocks is one contiguous 1 KiB store per wave.
//
// Compile with -ffp-contract=off (see dct.hpp).
#pragma clang fp contract(off)

#include "dct.hpp"
#include "kernels.hpp"

namespace jpeg_amd {

namespace {

constexpr int kThreads = 256;

// ---------------------------------------------------------------------------------------
// modulated table -> LDS (decode.swift:3984-4017); 64 lanes, one entry each
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ void modulate_to_lds(float *sq, const uint16_t *quanta, float scale)
{
    const int t = threadIdx.x;
    if (t < 64) {
        const int k = t & 7, h = t >> 3;
        sq[t] = modulate_entry(k, h, scale, quanta[zigzag_of(k, h)]);
    }
    __syncthreads();
}

// ---------------------------------------------------------------------------------------
// a3..a7  Spectral.Plane.idct(quanta:precision:)   decode.swift:4101-4133
// ---------------------------------------------------------------------------------------
template <typename OutT>
__global__ __launch_bounds__(kThreads) void k_idct_plane(
    const int16_t *__restrict__ coef, size_t coef_stride, const uint16_t *__restrict__ quanta,
    size_t quanta_stride, int qi, int ux, int nblocks, float level, float limit,
    OutT *__restrict__ out, size_t out_stride)
{
    __shared__ float sq[64];
    const int img = blockIdx.y;
    modulate_to_lds(sq, quanta + img * quanta_stride + 64 * qi, 0.125f);

    const int b = blockIdx.x * kThreads + threadIdx.x;
    if (b >= nblocks) return;
    const int by = b / ux, bx = b - by * ux;

    const uint4 *src = reinterpret_cast<const uint4 *>(coef + img * coef_stride + (size_t)64 * b);
    uint32_t w[32];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const uint4 v = src[i];
        w[4 * i + 0] = v.x; w[4 * i + 1] = v.y; w[4 * i + 2] = v.z; w[4 * i + 3] = v.w;
    }

    float g[64];
    idct_block(w, sq, level, g);

    const size_t pitch = (size_t)8 * ux;
    OutT *dst = out + img * out_stride + (size_t)8 * by * pitch + 8 * bx;
#pragma unroll
    for (int y = 0; y < 8; ++y) {
        uint32_t s[8];
#pragma unroll
        for (int x = 0; x < 8; ++x) s[x] = clamp_trunc(g[8 * y + x], limit);
        if constexpr (sizeof(OutT) == 2) {
            uint4 v;
            v.x = s[0] | (s[1] << 16); v.y = s[2] | (s[3] << 16);
            v.z = s[4] | (s[5] << 16); v.w = s[6] | (s[7] << 16);
            *reinterpret_cast<uint4 *>(dst + y * pitch) = v;
        } else {
            uint2 v;
            v.x = s[0] | (s[1] << 8) | (s[2] << 16) | (s[3] << 24);
            v.y = s[4] | (s[5] << 8) | (s[6] << 16) | (s[7] << 24);
            *reinterpret_cast<uint2 *>(dst + y * pitch) = v;
        }
    }
}

// ---------------------------------------------------------------------------------------
// colour math  jpeg.swift:441-453 (YCbCr.rgb), :463-478 (RGB.ycc), :343-354 (clamp)
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t clamp_u8(float v)
{
    return (uint32_t)__builtin_amdgcn_fmed3f(v, 0.0f, 255.0f);
}

// x = (Float(y) + m_cb * (Float(cb) - 128)) + m_cr * (Float(cr) - 128); the two `0.0 * c`
// products only add a signed zero, which cannot change any sum here.
__device__ __forceinline__ void ycc_to_rgb(float y, float cb, float cr, uint32_t &r,
                                           uint32_t &g, uint32_t &b)
{
    const float pb = cb - 128.0f;
    const float pr = cr - 128.0f;
    r = clamp_u8(y + 1.40200f * pr);
    g = clamp_u8((y + -0.34414f * pb) + -0.71414f * pr);
    b = clamp_u8(y + 1.77200f * pb);
}

// x = ((m0 + m_r * r) + m_g * g) + m_b * b; for Y m0 = 0 and `0 + x` is exact.
__device__ __forceinline__ uint32_t rgb_to_ycc_component(int p, float r, float g, float b)
{
    if (p == 0) return clamp_u8((0.2990f * r + 0.5870f * g) + 0.1140f * b);
    if (p == 1) return clamp_u8(((128.0f + -0.1687f * r) + -0.3313f * g) + 0.5000f * b);
    return clamp_u8(((128.0f + 0.5000f * r) + -0.4187f * g) + -0.0813f * b);
}

// ---------------------------------------------------------------------------------------
// a9 (+a11/a12)  Planar.interleaved(cosite:)  decode.swift:4182-4276, general form
// ---------------------------------------------------------------------------------------
struct UpsampleArgs {
    const void *plane[JPEG_AMD_MAX_PLANES];
    size_t stride[JPEG_AMD_MAX_PLANES];  // elements between images
    int pw[JPEG_AMD_MAX_PLANES], ph[JPEG_AMD_MAX_PLANES];
    int ax[JPEG_AMD_MAX_PLANES], bx[JPEG_AMD_MAX_PLANES], cx[JPEG_AMD_MAX_PLANES];
    int ay[JPEG_AMD_MAX_PLANES], by[JPEG_AMD_MAX_PLANES], cy[JPEG_AMD_MAX_PLANES];
    int direct[JPEG_AMD_MAX_PLANES];  // factor == scale, or single-plane image: crop copy
    int count, W, H;
};

template <typename T>
__device__ __forceinline__ uint32_t upsampled(const UpsampleArgs &a, int p, int img, int x, int y)
{
    const T *plane = static_cast<const T *>(a.plane[p]) + img * a.stride[p];
    const size_t pw = a.pw[p];
    if (a.direct[p]) return plane[x + pw * y];  // :4192, :4212

    // :4240-4241  quotientAndRemainder truncates toward zero, like C's / and %
    const int nx = a.ax[p] + a.bx[p] * x, ny = a.ay[p] + a.by[p] * y;
    const int ix = nx / a.cx[p], fx = nx - ix * a.cx[p];
    const int iy = ny / a.cy[p], fy = ny - iy * a.cy[p];
    const int jx = min(ix + 1, a.pw[p] - 1);  // :4245-4246 clamps to the PADDED plane
    const int jy = min(iy + 1, a.ph[p] - 1);
    const float tx = fmaxf(0.0f, fminf((float)fx / (float)a.cx[p], 1.0f));  // :4250-4251
    const float ty = fmaxf(0.0f, fminf((float)fy / (float)a.cy[p], 1.0f));
    const float u00 = (float)plane[ix + pw * iy], u01 = (float)plane[jx + pw * iy];
    const float u10 = (float)plane[ix + pw * jy], u11 = (float)plane[jx + pw * jy];
    const float v0 = u00 * (1.0f - tx) + u01 * tx;  // :4260-4261
    const float v1 = u10 * (1.0f - tx) + u11 * tx;
    return (uint32_t)round_half_away(v0 * (1.0f - ty) + v1 * ty);  // :4264
}

template <typename T, PixelKind KIND>
__global__ __launch_bounds__(kThreads) void k_planar_to_pixels(UpsampleArgs a, void *out,
                                                               size_t out_stride_bytes)
{
    const int img = blockIdx.y;
    const size_t i = (size_t)blockIdx.x * kThreads + threadIdx.x;
    if (i >= (size_t)a.W * a.H) return;
    const int y = (int)(i / a.W), x = (int)(i - (size_t)y * a.W);

    uint32_t s[JPEG_AMD_MAX_PLANES];
#pragma unroll
    for (int p = 0; p < JPEG_AMD_MAX_PLANES; ++p)
        if (p < a.count) s[p] = upsampled<T>(a, p, img, x, y);

    uint8_t *base = static_cast<uint8_t *>(out) + img * out_stride_bytes;
    if constexpr (KIND == PixelKind::Rect16) {
        uint16_t *o = reinterpret_cast<uint16_t *>(base) + i * a.count;
#pragma unroll
        for (int p = 0; p < JPEG_AMD_MAX_PLANES; ++p)
            if (p < a.count) o[p] = (uint16_t)s[p];
    } else {
        uint8_t *o = base + 3 * i;
        // jpeg.swift:499-503, 557-561: a grey image is (y, 128, 128)
        const uint32_t yy = s[0], cb = a.count == 1 ? 128u : s[1], cr = a.count == 1 ? 128u : s[2];
        if constexpr (KIND == PixelKind::YCC8) {
            o[0] = (uint8_t)yy; o[1] = (uint8_t)cb; o[2] = (uint8_t)cr;
        } else {
            uint32_t r, g, b;
            ycc_to_rgb((float)yy, (float)cb, (float)cr, r, g, b);
            o[0] = (uint8_t)r; o[1] = (uint8_t)g; o[2] = (uint8_t)b;
        }
    }
}

// ---------------------------------------------------------------------------------------
// a10..a12  Rectangular.unpack(as:)   jpeg.swift:493-514, 551-572
// ---------------------------------------------------------------------------------------
template <bool RGB>
__global__ __launch_bounds__(kThreads) void k_unpack(const uint16_t *__restrict__ rect,
                                                     size_t npixels, int nplanes,
                                                     uint8_t *__restrict__ out)
{
    const size_t i = (size_t)blockIdx.x * kThreads + threadIdx.x;
    if (i >= npixels) return;
    uint32_t y, cb, cr;
    if (nplanes == 1) {
        y = rect[i]; cb = 128; cr = 128;
    } else {
        y = rect[3 * i]; cb = rect[3 * i + 1]; cr = rect[3 * i + 2];
    }
    // UInt8(UInt16) in the reference traps above 255 (unreachable after idct's clamp);
    // here the value is narrowed like the YCbCr(y:cb:cr:) initialiser would store it.
    y &= 0xff; cb &= 0xff; cr &= 0xff;
    uint8_t *o = out + 3 * i;
    if constexpr (RGB) {
        uint32_t r, g, b;
        ycc_to_rgb((float)y, (float)cb, (float)cr, r, g, b);
        o[0] = (uint8_t)r; o[1] = (uint8_t)g; o[2] = (uint8_t)b;
    } else {
        o[0] = (uint8_t)y; o[1] = (uint8_t)cb; o[2] = (uint8_t)cr;
    }
}

// ---------------------------------------------------------------------------------------
// a13  Rectangular.pack   jpeg.swift:527-539, 584-599
// ---------------------------------------------------------------------------------------
template <bool RGB>
__global__ __launch_bounds__(kThreads) void k_pack(const uint8_t *__restrict__ px,
                                                   size_t npixels, int nplanes,
                                                   uint16_t *__restrict__ rect)
{
    const size_t i = (size_t)blockIdx.x * kThreads + threadIdx.x;
    if (i >= npixels) return;
    const uint32_t c0 = px[3 * i], c1 = px[3 * i + 1], c2 = px[3 * i + 2];
    uint32_t y, cb, cr;
    if constexpr (RGB) {
        const float r = (float)c0, g = (float)c1, b = (float)c2;
        y  = rgb_to_ycc_component(0, r, g, b);
        cb = rgb_to_ycc_component(1, r, g, b);
        cr = rgb_to_ycc_component(2, r, g, b);
    } else {
        y = c0; cb = c1; cr = c2;
    }
    if (nplanes == 1) {
        rect[i] = (uint16_t)y;
    } else {
        rect[3 * i] = (uint16_t)y; rect[3 * i + 1] = (uint16_t)cb; rect[3 * i + 2] = (uint16_t)cr;
    }
}

// ---------------------------------------------------------------------------------------
// a14  Rectangular.decomposed()   encode.swift:389-425
// ---------------------------------------------------------------------------------------
struct DecomposeArgs {
    const void *in;
    size_t in_stride_bytes;
    void *plane;
    size_t plane_stride;  // elements
    int W, H, count, p;
    int fx, fy, sx, sy;   // factor, scale
    int pw, ph;           // padded plane size = 8 * units
};

template <PixelKind KIND>
__device__ __forceinline__ uint32_t source_sample(const DecomposeArgs &a, const uint8_t *base,
                                                  int x, int y)
{
    const size_t i = (size_t)a.W * y + x;
    if constexpr (KIND == PixelKind::Rect16) {
        return reinterpret_cast<const uint16_t *>(base)[i * a.count + a.p];
    } else if constexpr (KIND == PixelKind::YCC8) {
        return base[3 * i + a.p];
    } else {
        const float r = (float)base[3 * i], g = (float)base[3 * i + 1], b = (float)base[3 * i + 2];
        return rgb_to_ycc_component(a.p, r, g, b);
    }
}

template <PixelKind KIND>
__global__ __launch_bounds__(kThreads) void k_decompose(DecomposeArgs a)
{
    const int img = blockIdx.y;
    const size_t i = (size_t)blockIdx.x * kThreads + threadIdx.x;
    if (i >= (size_t)a.pw * a.ph) return;
    const int y = (int)(i / a.pw), x = (int)(i - (size_t)y * a.pw);
    const uint8_t *base = static_cast<const uint8_t *>(a.in) + img * a.in_stride_bytes;

    const int rx = a.sx / a.fx, ry = a.sy / a.fy;            // response     :403
    const int bx = x * a.sx / a.fx, by = y * a.sy / a.fy;    // base         :407-411
    uint32_t sum = 0;
    for (int yy = by; yy < by + ry; ++yy) {
        const int iy = min(yy, a.H - 1);                     // edge replicate :415-417
        for (int xx = bx; xx < bx + rx; ++xx) sum += source_sample<KIND>(a, base, min(xx, a.W - 1), iy);
    }
    const float magnitude = (float)(rx * ry);                // :404
    uint16_t *plane = static_cast<uint16_t *>(a.plane) + img * a.plane_stride;
    plane[i] = (uint16_t)(uint32_t)((float)sum / magnitude); // :422 truncating
}

// ---------------------------------------------------------------------------------------
// a15..a17  Spectral.Plane.fdct(_:quanta:precision:)   encode.swift:199-248
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(kThreads) void k_fdct_plane(
    const uint16_t *__restrict__ plane, size_t plane_stride, const uint16_t *__restrict__ quanta,
    size_t quanta_stride, int qi, int ux, int nblocks, float level, float limit,
    int16_t *__restrict__ coef, size_t coef_stride)
{
    __shared__ float sq[64];
    const int img = blockIdx.y;
    modulate_to_lds(sq, quanta + img * quanta_stride + 64 * qi, 8.0f);  // :205-209

    const int b = blockIdx.x * kThreads + threadIdx.x;
    if (b >= nblocks) return;
    const int by = b / ux, bx = b - by * ux;
    const size_t pitch = (size_t)8 * ux;
    const uint16_t *src = plane + img * plane_stride + (size_t)8 * by * pitch + 8 * bx;

    float g[64];
#pragma unroll
    for (int y = 0; y < 8; ++y) {
        const uint4 v = *reinterpret_cast<const uint4 *>(src + y * pitch);
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int x = 0; x < 8; ++x) {
            const float s = (float)((w[x >> 1] >> (16 * (x & 1))) & 0xffffu);
            g[8 * y + x] = fminf(limit, s);  // pointwiseMin(limit, .)  encode.swift:85
        }
    }

    float H[64];
    fdct_block(g, level, H);

    // quantise (true division, round half away) and scatter to zigzag order :225-240
    uint32_t w[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) w[i] = 0;
#pragma unroll
    for (int h = 0; h < 8; ++h) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const float v = H[8 * h + k] / sq[8 * h + k];
            const int32_t c = (int32_t)round_half_away(v);
            const int z = zigzag_of(k, h);
            w[z >> 1] |= ((uint32_t)c & 0xffffu) << (16 * (z & 1));
        }
    }
    uint4 *dst = reinterpret_cast<uint4 *>(coef + img * coef_stride + (size_t)64 * b);
#pragma unroll
    for (int i = 0; i < 8; ++i) dst[i] = make_uint4(w[4 * i], w[4 * i + 1], w[4 * i + 2], w[4 * i + 3]);
}

inline unsigned blocks_for(size_t n) { return (unsigned)((n + kThreads - 1) / kThreads); }

// ---- planes -> sparse coefficients (the device side of jpeg_amd_jpeg_encode_sparse, entropy_encode.cpp) --------------------
// One block per work-item: its nonzero coefficients -- and the DC in any case -- become entries (value, zigzag index, last-of-
// block flag), the entries of a workgroup's 256 blocks one run in the image's arena (space drawn from the image's cursor with
// one atomic per workgroup; the order of the runs is that of the workgroups' arrival -- the descriptors say where a block's
// entries are).  An image whose entries do not fit `capacity` ends with cursor > capacity: its planes have to come down whole.
struct SparsifyArgs {
    const int16_t *coef[JPEG_AMD_MAX_PLANES];
    size_t coef_stride[JPEG_AMD_MAX_PLANES];
    uint32_t first[JPEG_AMD_MAX_PLANES + 1];
    int nplanes;
    uint32_t *desc;      size_t desc_stride;
    uint32_t *entries;   size_t entries_stride;
    uint32_t capacity;
    uint32_t *cursor;    // per image, zero before the launch
};

__global__ __launch_bounds__(kThreads) void k_sparsify(SparsifyArgs a)
{
    __shared__ uint32_t wave_total[kThreads / 64];
    __shared__ uint32_t group_base;
    const int img = blockIdx.y, t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const uint32_t b = blockIdx.x * kThreads + t;
    const bool live = b < a.first[a.nplanes];
    uint32_t w[32];
    uint32_t count = 0;
    if (live) {
        int p = 0;
#pragma unroll
        for (int q = 1; q < JPEG_AMD_MAX_PLANES; ++q) p += (q < a.nplanes && b >= a.first[q]);
        const int16_t *src = a.coef[0];
        size_t stride = a.coef_stride[0];
        uint32_t first = a.first[0];
#pragma unroll
        for (int q = 1; q < JPEG_AMD_MAX_PLANES; ++q)
            if (p == q) { src = a.coef[q]; stride = a.coef_stride[q]; first = a.first[q]; }
        const uint4 *blk = reinterpret_cast<const uint4 *>(src + img * stride + (size_t)(b - first) * 64);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const uint4 v = blk[i];
            w[4 * i] = v.x; w[4 * i + 1] = v.y; w[4 * i + 2] = v.z; w[4 * i + 3] = v.w;
        }
        count = 1;                                              // the DC, zero or not
        if (w[0] >> 16) ++count;
#pragma unroll
        for (int i = 1; i < 32; ++i) count += ((w[i] & 0xffffu) != 0) + ((w[i] >> 16) != 0);
    }
    // exclusive prefix of `count` over the workgroup: within the wave by shuffles, across the four waves through LDS
    uint32_t incl = count;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t up = __shfl_up(incl, d, 64);
        if (lane >= d) incl += up;
    }
    if (lane == 63) wave_total[wave] = incl;
    __syncthreads();
    uint32_t before = 0, total = 0;
#pragma unroll
    for (int k = 0; k < kThreads / 64; ++k) { if (k < wave) before += wave_total[k]; total += wave_total[k]; }
    if (t == 0) group_base = atomicAdd(a.cursor + img, total);
    __syncthreads();
    const uint32_t base = group_base;
    if (!live || base + total > a.capacity) return;             // (every workgroup of an overflowing image may or may not write: the host looks at the cursor)
    uint32_t at = base + before + incl - count;
    a.desc[img * a.desc_stride + b] = at;
    uint32_t *e = a.entries + img * a.entries_stride;
    uint32_t left = count;
#pragma unroll
    for (int i = 0; i < 32; ++i) {
        const uint32_t lo = w[i] & 0xffffu, hi = w[i] >> 16;
        if (lo != 0 || i == 0) { --left; e[at++] = lo | (uint32_t)(2 * i) << 16 | (left == 0 ? 0x80000000u : 0u); }
        if (hi != 0) { --left; e[at++] = hi | (uint32_t)(2 * i + 1) << 16 | (left == 0 ? 0x80000000u : 0u); }
    }
}

// ---- sparse coefficients -> planes (the device side of jpeg_amd_jpeg_decode_sparse, entropy.cpp) -------------------------
// Eight work-items per block, one 16-byte octet of the block each: they walk the block's entries together (the same address
// for all eight: one fetch) and keep what falls into their octet; a workgroup writes 32 whole blocks = 4 KiB in a row.
// Blocks no scan reached (descriptor 0xFFFFFFFF) and images flagged in `skip` (their planes were uploaded as they are) aside,
// every byte of the planes is written: no memset in front.
struct ExpandArgs {
    const uint32_t *desc;      size_t desc_stride;       // per image: one descriptor per block, planes in frame order
    const uint32_t *entries;   size_t entries_stride;    // per image: the entry arena
    const uint8_t *skip;                                 // optional, per image
    const uint64_t *packed;                              // optional, per image: where in `desc` its record [descriptors][entries] begins (elements)
    int16_t *coef[JPEG_AMD_MAX_PLANES];
    size_t coef_stride[JPEG_AMD_MAX_PLANES];
    uint32_t first[JPEG_AMD_MAX_PLANES + 1];             // first[p]: blocks of the planes before p
    int nplanes;
};

__global__ __launch_bounds__(kThreads) void k_expand_sparse(ExpandArgs a)
{
    const int img = blockIdx.y;
    if (a.skip && a.skip[img]) return;
    const uint32_t b = blockIdx.x * (kThreads / 8) + (threadIdx.x >> 3);
    if (b >= a.first[a.nplanes]) return;
    const uint32_t octet = threadIdx.x & 7;
    int p = 0;
#pragma unroll
    for (int q = 1; q < JPEG_AMD_MAX_PLANES; ++q) p += (q < a.nplanes && b >= a.first[q]);
    uint32_t w[4] = {0, 0, 0, 0};
    const uint32_t *desc = a.packed ? a.desc + a.packed[img] : a.desc + img * a.desc_stride;
    uint32_t at = desc[b];
    if (at != 0xffffffffu) {
        const uint32_t *e = a.packed ? desc + a.first[a.nplanes] : a.entries + img * a.entries_stride;
        for (;; ++at) {
            const uint32_t v = e[at];
            const uint32_t pos = (v >> 16) & 63;
            if ((pos >> 3) == octet) {
                const uint32_t half = (pos & 1) * 16, word = (pos >> 1) & 3;
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (word == (uint32_t)k) w[k] = (w[k] & ~(0xffffu << half)) | ((v & 0xffffu) << half);
            }
            if (v >> 31) break;
        }
    }
    int16_t *dst = a.coef[0];
    size_t stride = a.coef_stride[0];
    uint32_t first = a.first[0];
#pragma unroll
    for (int q = 1; q < JPEG_AMD_MAX_PLANES; ++q)
        if (p == q) { dst = a.coef[q]; stride = a.coef_stride[q]; first = a.first[q]; }
    *reinterpret_cast<uint4 *>(dst + img * stride + (size_t)(b - first) * 64 + 8 * octet) = make_uint4(w[0], w[1], w[2], w[3]);
}

}  // namespace

// =======================================================================================
// launchers
// =======================================================================================

hipError_t launch_sparsify(hipStream_t stream, int n_images, const jpeg_amd_layout &L, const PlaneSet &coef, uint32_t *d_desc,
                           size_t desc_stride, uint32_t *d_entries, size_t entries_stride, uint32_t capacity, uint32_t *d_cursor)
{
    SparsifyArgs a{};
    a.desc = d_desc; a.desc_stride = desc_stride; a.entries = d_entries; a.entries_stride = entries_stride;
    a.capacity = capacity; a.cursor = d_cursor; a.nplanes = L.nplanes;
    uint32_t blocks = 0;
    for (int p = 0; p < L.nplanes; ++p) {
        a.coef[p] = static_cast<const int16_t *>(coef.ptr[p]); a.coef_stride[p] = coef.stride[p];
        a.first[p] = blocks;
        blocks += (uint32_t)L.units_x[p] * (uint32_t)L.units_y[p];
    }
    for (int p = L.nplanes; p <= JPEG_AMD_MAX_PLANES; ++p) a.first[p] = blocks;
    if (blocks == 0 || n_images == 0) return hipSuccess;
    const hipError_t e = hipMemsetAsync(d_cursor, 0, (size_t)n_images * sizeof(uint32_t), stream);
    if (e != hipSuccess) return e;
    const dim3 grid((blocks + kThreads - 1) / kThreads, n_images);
    hipLaunchKernelGGL(k_sparsify, grid, dim3(kThreads), 0, stream, a);
    return hipGetLastError();
}

hipError_t launch_expand_sparse(hipStream_t stream, int n_images, const jpeg_amd_layout &L, const uint32_t *d_desc, size_t desc_stride,
                                const uint32_t *d_entries, size_t entries_stride, const uint8_t *d_skip, const PlaneSetMut &coef,
                                const uint64_t *d_packed)
{
    ExpandArgs a{};
    a.desc = d_desc; a.desc_stride = desc_stride; a.entries = d_entries; a.entries_stride = entries_stride; a.skip = d_skip;
    a.packed = d_packed;
    a.nplanes = L.nplanes;
    uint32_t blocks = 0;
    for (int p = 0; p < L.nplanes; ++p) {
        a.coef[p] = static_cast<int16_t *>(coef.ptr[p]); a.coef_stride[p] = coef.stride[p];
        a.first[p] = blocks;
        blocks += (uint32_t)L.units_x[p] * (uint32_t)L.units_y[p];
    }
    for (int p = L.nplanes; p <= JPEG_AMD_MAX_PLANES; ++p) a.first[p] = blocks;
    if (blocks == 0 || n_images == 0) return hipSuccess;
    const dim3 grid((blocks + kThreads / 8 - 1) / (kThreads / 8), n_images);
    hipLaunchKernelGGL(k_expand_sparse, grid, dim3(kThreads), 0, stream, a);
    return hipGetLastError();
}

hipError_t launch_idct_plane(hipStream_t stream, int n_images, const int16_t *d_coef,
                             size_t coef_stride, QuantaRef q, int qi, int ux, int uy,
                             int precision, void *d_plane, size_t plane_stride, bool out_u8)
{
    const int nblocks = ux * uy;
    if (nblocks == 0 || n_images == 0) return hipSuccess;
    // decode.swift:4110-4113
    const float level = ldexpf(1.0f, precision - 1) + 0.5f;
    const float limit = ldexpf(1.0f, precision) - 1.0f;
    const dim3 grid(blocks_for(nblocks), n_images);
    if (out_u8)
        hipLaunchKernelGGL(k_idct_plane<uint8_t>, grid, dim3(kThreads), 0, stream, d_coef,
                           coef_stride, q.d_quanta, q.image_stride, qi, ux, nblocks, level,
                           limit, static_cast<uint8_t *>(d_plane), plane_stride);
    else
        hipLaunchKernelGGL(k_idct_plane<uint16_t>, grid, dim3(kThreads), 0, stream, d_coef,
                           coef_stride, q.d_quanta, q.image_stride, qi, ux, nblocks, level,
                           limit, static_cast<uint16_t *>(d_plane), plane_stride);
    return hipGetLastError();
}

hipError_t launch_planar_to_pixels(hipStream_t stream, int n_images,
                                   const jpeg_amd_layout &L, const PlaneSet &planes,
                                   bool planes_u8, bool cosited, PixelKind kind, void *d_out,
                                   size_t out_stride_bytes)
{
    UpsampleArgs a{};
    a.count = L.nplanes; a.W = L.width; a.H = L.height;
    for (int p = 0; p < L.nplanes; ++p) {
        a.plane[p] = planes.ptr[p];
        a.stride[p] = planes.stride[p];
        a.pw[p] = 8 * L.units_x[p];
        a.ph[p] = 8 * L.units_y[p];
        a.direct[p] = (L.nplanes == 1) ||
                      (L.factor_x[p] == L.scale_x && L.factor_y[p] == L.scale_y);
        if (cosited) {  // decode.swift:4223-4234
            a.ax[p] = 0; a.ay[p] = 0;
            a.bx[p] = L.factor_x[p]; a.by[p] = L.factor_y[p];
            a.cx[p] = L.scale_x;     a.cy[p] = L.scale_y;
        } else {
            a.ax[p] = L.factor_x[p] - L.scale_x; a.ay[p] = L.factor_y[p] - L.scale_y;
            a.bx[p] = 2 * L.factor_x[p];         a.by[p] = 2 * L.factor_y[p];
            a.cx[p] = 2 * L.scale_x;             a.cy[p] = 2 * L.scale_y;
        }
    }
    const dim3 grid(blocks_for((size_t)L.width * L.height), n_images);
#define JA_LAUNCH(T, K) \
    hipLaunchKernelGGL((k_planar_to_pixels<T, K>), grid, dim3(kThreads), 0, stream, a, d_out, out_stride_bytes)
    if (planes_u8) {
        if (kind == PixelKind::Rect16) JA_LAUNCH(uint8_t, PixelKind::Rect16);
        else if (kind == PixelKind::YCC8) JA_LAUNCH(uint8_t, PixelKind::YCC8);
        else JA_LAUNCH(uint8_t, PixelKind::RGB8);
    } else {
        if (kind == PixelKind::Rect16) JA_LAUNCH(uint16_t, PixelKind::Rect16);
        else if (kind == PixelKind::YCC8) JA_LAUNCH(uint16_t, PixelKind::YCC8);
        else JA_LAUNCH(uint16_t, PixelKind::RGB8);
    }
#undef JA_LAUNCH
    return hipGetLastError();
}

hipError_t launch_unpack(hipStream_t stream, const uint16_t *d_rect, size_t npixels,
                         int nplanes, jpeg_amd_color color, uint8_t *d_pixels)
{
    if (npixels == 0) return hipSuccess;
    const dim3 grid(blocks_for(npixels));
    if (color == JPEG_AMD_COLOR_RGB8)
        hipLaunchKernelGGL(k_unpack<true>, grid, dim3(kThreads), 0, stream, d_rect, npixels, nplanes, d_pixels);
    else
        hipLaunchKernelGGL(k_unpack<false>, grid, dim3(kThreads), 0, stream, d_rect, npixels, nplanes, d_pixels);
    return hipGetLastError();
}

hipError_t launch_pack(hipStream_t stream, const uint8_t *d_pixels, size_t npixels,
                       int nplanes, jpeg_amd_color color, uint16_t *d_rect)
{
    if (npixels == 0) return hipSuccess;
    const dim3 grid(blocks_for(npixels));
    if (color == JPEG_AMD_COLOR_RGB8)
        hipLaunchKernelGGL(k_pack<true>, grid, dim3(kThreads), 0, stream, d_pixels, npixels, nplanes, d_rect);
    else
        hipLaunchKernelGGL(k_pack<false>, grid, dim3(kThreads), 0, stream, d_pixels, npixels, nplanes, d_rect);
    return hipGetLastError();
}

hipError_t launch_decompose(hipStream_t stream, int n_images, const jpeg_amd_layout &L,
                            const void *d_in, size_t in_stride_bytes, PixelKind in_kind,
                            const PlaneSetMut &planes)
{
    for (int p = 0; p < L.nplanes; ++p) {
        DecomposeArgs a{};
        a.in = d_in; a.in_stride_bytes = in_stride_bytes;
        a.plane = planes.ptr[p]; a.plane_stride = planes.stride[p];
        a.W = L.width; a.H = L.height; a.count = L.nplanes; a.p = p;
        a.fx = L.factor_x[p]; a.fy = L.factor_y[p]; a.sx = L.scale_x; a.sy = L.scale_y;
        a.pw = 8 * L.units_x[p]; a.ph = 8 * L.units_y[p];
        if (a.pw == 0 || a.ph == 0) continue;
        const dim3 grid(blocks_for((size_t)a.pw * a.ph), n_images);
        if (in_kind == PixelKind::Rect16)
            hipLaunchKernelGGL(k_decompose<PixelKind::Rect16>, grid, dim3(kThreads), 0, stream, a);
        else if (in_kind == PixelKind::YCC8)
            hipLaunchKernelGGL(k_decompose<PixelKind::YCC8>, grid, dim3(kThreads), 0, stream, a);
        else
            hipLaunchKernelGGL(k_decompose<PixelKind::RGB8>, grid, dim3(kThreads), 0, stream, a);
        const hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

hipError_t launch_fdct_plane(hipStream_t stream, int n_images, const uint16_t *d_plane,
                             size_t plane_stride, QuantaRef q, int qi, int ux, int uy,
                             int precision, int16_t *d_coef, size_t coef_stride)
{
    const int nblocks = ux * uy;
    if (nblocks == 0 || n_images == 0) return hipSuccess;
    // encode.swift:215-218: no +0.5 in the forward level shift
    const float level = ldexpf(1.0f, precision - 1) * 8.0f;
    const float limit = ldexpf(1.0f, precision) - 1.0f;
    const dim3 grid(blocks_for(nblocks), n_images);
    hipLaunchKernelGGL(k_fdct_plane, grid, dim3(kThreads), 0, stream, d_plane, plane_stride,
                       q.d_quanta, q.image_stride, qi, ux, nblocks, level, limit, d_coef,
                       coef_stride);
    return hipGetLastError();
}

}  // namespace jpeg_amd
