// probe_typed.hip -- round 6: can the texture path and a few cheaper opcodes take conversions off the VALU of k_encode_fused?
//   T1  tbuffer_load_format_xyzw, 8_8_8_8 USCALED: four bytes -> four floats in the memory pipeline (exact? at which byte alignments?)
//   T2  tbuffer_store_format_xyzw, 8_8_8_8 USCALED / UINT and 16_16_16_16 SSCALED / SINT: how does the store convert a float (rounding, clamp)?
//   T3  v_cvt_i32_f32_sdwa dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE: the second coefficient of a zigzag pair converted INTO the upper half
//       of the register that already holds the first (instead of v_cvt_i32 + v_cvt_pk_i16_i32)
//   T4  copysign as ONE v_bitop3_b32 (full rate) instead of v_bfi_b32 (half rate)
//   T5  the encode kernel's pixel fetch alone (a 4096 x 4096 RGB8 frame, one 8 x 8 block per work-item, 8 rows x 24 bytes):
//       (A) 3 x uint2 per row + 24 v_cvt_f32_ubyte   (B) 6 x tbuffer_load_format_xyzw per row -- time per pass, and a checksum
// build: hipcc --offload-arch=gfx950 -O3 tools/probe_typed.hip -o tools/probe_typed
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <cmath>
#include <vector>
#include <algorithm>

typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ i32x4 make_srd(const void *p, uint32_t bytes)
{
    const uint64_t a = (uint64_t)p;
    i32x4 r;
    r.x = __builtin_amdgcn_readfirstlane((int)(uint32_t)a);
    r.y = __builtin_amdgcn_readfirstlane((int)(uint32_t)(a >> 32) & 0xffff);
    r.z = __builtin_amdgcn_readfirstlane((int)bytes);
    r.w = 0x00020000;
    return r;
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

// ---- T1 ----
__global__ void t1_load(const uint8_t *src, uint32_t bytes, uint32_t shift, float *out)
{
    const i32x4 srd = make_srd(src, bytes);
    const uint32_t voff = shift + 24 * threadIdx.x;
    f32x4 a, b;
    asm volatile("s_nop 4\n\t"
                 "tbuffer_load_format_xyzw %0, %2, %3, 0 format:[BUF_DATA_FORMAT_8_8_8_8,BUF_NUM_FORMAT_USCALED] offen\n\t"
                 "tbuffer_load_format_xyzw %1, %2, %3, 0 format:[BUF_DATA_FORMAT_8_8_8_8,BUF_NUM_FORMAT_USCALED] offen offset:20\n\t"
                 "s_waitcnt vmcnt(0)" : "=&v"(a), "=&v"(b) : "v"(voff), "s"(srd) : "memory");
    float *o = out + 8 * threadIdx.x;
    o[0] = a.x; o[1] = a.y; o[2] = a.z; o[3] = a.w; o[4] = b.x; o[5] = b.y; o[6] = b.z; o[7] = b.w;
}

// ---- T2 ----
template <int FMT>
__global__ void t2_store(uint8_t *dst, uint32_t bytes, const float *vals)
{
    const i32x4 srd = make_srd(dst, bytes);
    const f32x4 v = {vals[4 * threadIdx.x], vals[4 * threadIdx.x + 1], vals[4 * threadIdx.x + 2], vals[4 * threadIdx.x + 3]};
    if constexpr (FMT == 0) {
        const uint32_t voff = 4 * threadIdx.x;
        asm volatile("s_nop 4\n\ttbuffer_store_format_xyzw %0, %1, %2, 0 format:[BUF_DATA_FORMAT_8_8_8_8,BUF_NUM_FORMAT_USCALED] offen\n\ts_waitcnt vmcnt(0)" ::"v"(v), "v"(voff), "s"(srd) : "memory");
    } else if constexpr (FMT == 1) {
        const uint32_t voff = 4 * threadIdx.x;
        asm volatile("s_nop 4\n\ttbuffer_store_format_xyzw %0, %1, %2, 0 format:[BUF_DATA_FORMAT_8_8_8_8,BUF_NUM_FORMAT_UINT] offen\n\ts_waitcnt vmcnt(0)" ::"v"(v), "v"(voff), "s"(srd) : "memory");
    } else if constexpr (FMT == 2) {
        const uint32_t voff = 8 * threadIdx.x;
        asm volatile("s_nop 4\n\ttbuffer_store_format_xyzw %0, %1, %2, 0 format:[BUF_DATA_FORMAT_16_16_16_16,BUF_NUM_FORMAT_SSCALED] offen\n\ts_waitcnt vmcnt(0)" ::"v"(v), "v"(voff), "s"(srd) : "memory");
    } else {
        const uint32_t voff = 8 * threadIdx.x;
        asm volatile("s_nop 4\n\ttbuffer_store_format_xyzw %0, %1, %2, 0 format:[BUF_DATA_FORMAT_16_16_16_16,BUF_NUM_FORMAT_SINT] offen\n\ts_waitcnt vmcnt(0)" ::"v"(v), "v"(voff), "s"(srd) : "memory");
    }
}

// ---- T3 / T4 ----
__global__ void t3_pack(const float *lo, const float *hi, uint32_t *out, uint32_t *out_bitop, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float a = lo[i], b = hi[i];
    uint32_t d;
    asm volatile("v_cvt_i32_f32_e32 %0, %1\n\t"
                 "s_nop 0\n\t"
                 "v_cvt_i32_f32_sdwa %0, %2 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD" : "=&v"(d) : "v"(a), "v"(b));
    out[i] = d;
    // copysign(0.49999997f, a) by v_bitop3_b32: (magnitude & 0x7fffffff) | (a & 0x80000000)  ==  bitfield select with mask 0x7fffffff
    // truth table over (S0 = mask, S1 = magnitude, S2 = sign source): S0 ? S1 : S2  -> 0xca
    uint32_t c;
    asm volatile("v_bitop3_b32 %0, %1, %2, %3 bitop3:0xca" : "=v"(c) : "v"(0x7fffffffu), "v"(0.49999997f), "v"(a));
    out_bitop[i] = c;
}


// ---- T6: D16 typed load (four bytes -> four f16 in two registers) + v_fma_mix_f32 consuming the halves without a conversion ----
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
__global__ void t6_d16(const uint8_t *src, uint32_t bytes, float *out, uint32_t *raw)
{
    const i32x4 srd = make_srd(src, bytes);
    const uint32_t voff = 4 * threadIdx.x;
    u32x2 h;
    asm volatile("s_nop 4\n\t"
                 "tbuffer_load_format_d16_xyzw %0, %1, %2, 0 format:[BUF_DATA_FORMAT_8_8_8_8,BUF_NUM_FORMAT_USCALED] offen\n\t"
                 "s_waitcnt vmcnt(0)" : "=&v"(h) : "v"(voff), "s"(srd) : "memory");
    raw[2 * threadIdx.x] = h.x; raw[2 * threadIdx.x + 1] = h.y;
    const float c = 0.2990f, z = 0.0f;
    float p0, p1, p2, p3;
    // D = S0 * S1 + S2; op_sel_hi[i] = 1: source i is f16, op_sel[i] picks its half
    asm volatile("v_fma_mix_f32 %0, %4, %5, %7 op_sel_hi:[0,1,0]\n\t"
                 "v_fma_mix_f32 %1, %4, %5, %7 op_sel:[0,1,0] op_sel_hi:[0,1,0]\n\t"
                 "v_fma_mix_f32 %2, %4, %6, %7 op_sel_hi:[0,1,0]\n\t"
                 "v_fma_mix_f32 %3, %4, %6, %7 op_sel:[0,1,0] op_sel_hi:[0,1,0]"
                 : "=&v"(p0), "=&v"(p1), "=&v"(p2), "=&v"(p3) : "v"(c), "v"(h.x), "v"(h.y), "v"(z));
    out[4 * threadIdx.x] = p0; out[4 * threadIdx.x + 1] = p1; out[4 * threadIdx.x + 2] = p2; out[4 * threadIdx.x + 3] = p3;
}

// ---- T5 ----
template <int N>
__device__ __forceinline__ float ubyte(uint32_t v) { return (float)((v >> (8 * N)) & 0xffu); }

// (A) the product kernel's fetch: scalar row base + one per-lane offset, 3 x uint2 per row, 24 conversions per row
__global__ __launch_bounds__(256, 4) void t5_plain(const uint8_t *px, int W, int tiles_x, float *out)
{
    const int tyi = blockIdx.x / tiles_x, txi = blockIdx.x - tyi * tiles_x;
    const int lbx = threadIdx.x & 31, lby = threadIdx.x >> 5;
    const uint8_t *tile0 = px + ((size_t)(64 * tyi) * W + (size_t)256 * txi) * 3;
    const uint32_t voff = ((uint32_t)(8 * lby) * (uint32_t)W + 8u * lbx) * 3u;
    uint32_t pix[8][6];
#pragma unroll
    for (int y = 0; y < 8; ++y) {
        const uint2 *row = reinterpret_cast<const uint2 *>(tile0 + (size_t)y * W * 3 + voff);
        const uint2 p0 = row[0], p1 = row[1], p2 = row[2];
        pix[y][0] = p0.x; pix[y][1] = p0.y; pix[y][2] = p1.x; pix[y][3] = p1.y; pix[y][4] = p2.x; pix[y][5] = p2.y;
    }
    float acc[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int y = 0; y < 8; ++y) {
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int d = 0; d < 6; ++d) {
            const uint32_t dw = pix[y][d];
            acc[(4 * d + 0) % 3] += ubyte<0>(dw); acc[(4 * d + 1) % 3] += ubyte<1>(dw);
            acc[(4 * d + 2) % 3] += ubyte<2>(dw); acc[(4 * d + 3) % 3] += ubyte<3>(dw);
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc[0] + 2.0f * acc[1] + 4.0f * acc[2];
}

// (B) the same bytes through the texture path: 6 typed loads per row, floats arrive in the registers
template <int AHEAD>   // rows requested ahead of the row being consumed (8 = everything up front)
__global__ __launch_bounds__(256, 4) void t5_typed(const uint8_t *px, int W, int H, int tiles_x, float *out)
{
    const int tyi = blockIdx.x / tiles_x, txi = blockIdx.x - tyi * tiles_x;
    const int lbx = threadIdx.x & 31, lby = threadIdx.x >> 5;
    const uint8_t *tile0 = px + ((size_t)(64 * tyi) * W + (size_t)256 * txi) * 3;
    const i32x4 srd = make_srd(tile0, (uint32_t)std::min<size_t>((size_t)64 * W * 3, 0xffffffffull));
    const uint32_t voff = ((uint32_t)(8 * lby) * (uint32_t)W + 8u * lbx) * 3u;
    const uint32_t pitch = 3u * (uint32_t)W;
    f32x4 r[8][6];
    auto request = [&](int y) {
        const uint32_t soff = __builtin_amdgcn_readfirstlane((uint32_t)y * pitch);
#define TL(i, off) asm volatile("tbuffer_load_format_xyzw %0, %1, %2, %3 format:[BUF_DATA_FORMAT_8_8_8_8,BUF_NUM_FORMAT_USCALED] offen offset:" #off \
                                : "=v"(r[y][i]) : "v"(voff), "s"(srd), "s"(soff) : "memory")
        TL(0, 0); TL(1, 4); TL(2, 8); TL(3, 12); TL(4, 16); TL(5, 20);
#undef TL
    };
    float acc[3] = {0.f, 0.f, 0.f};
    if constexpr (AHEAD >= 8) {
#pragma unroll
        for (int y = 0; y < 8; ++y) request(y);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int y = 0; y < 8; ++y)
#pragma unroll
            for (int d = 0; d < 6; ++d) {
                asm volatile("" : "+v"(r[y][d]));
                acc[(4 * d + 0) % 3] += r[y][d].x; acc[(4 * d + 1) % 3] += r[y][d].y;
                acc[(4 * d + 2) % 3] += r[y][d].z; acc[(4 * d + 3) % 3] += r[y][d].w;
            }
    } else {
#pragma unroll
        for (int y = 0; y < AHEAD; ++y) request(y);
#pragma unroll
        for (int y = 0; y < 8; ++y) {
            if (y + AHEAD < 8) { request(y + AHEAD); asm volatile("s_waitcnt vmcnt(%0)" ::"n"(6 * AHEAD) : "memory"); }
            else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(6 * (7 - y)) : "memory");
#pragma unroll
            for (int d = 0; d < 6; ++d) {
                asm volatile("" : "+v"(r[y][d]));
                acc[(4 * d + 0) % 3] += r[y][d].x; acc[(4 * d + 1) % 3] += r[y][d].y;
                acc[(4 * d + 2) % 3] += r[y][d].z; acc[(4 * d + 3) % 3] += r[y][d].w;
            }
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc[0] + 2.0f * acc[1] + 4.0f * acc[2];
}

template <typename F>
static float time_us(F &&launch, int reps)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 5; ++i) launch(i);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < reps; ++i) launch(i);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e3f / reps;
}

int main()
{
    // ---- T1 ----
    {
        const uint32_t n = 64 * 24 + 64;
        std::vector<uint8_t> h(n);
        for (uint32_t i = 0; i < n; ++i) h[i] = (uint8_t)(i * 37u + (i >> 3) * 11u + 5u);
        uint8_t *d; float *o;
        CK(hipMalloc(&d, n)); CK(hipMalloc(&o, 64 * 8 * 4));
        CK(hipMemcpy(d, h.data(), n, hipMemcpyHostToDevice));
        for (uint32_t shift = 0; shift < 4; ++shift) {
            CK(hipMemset(o, 0xff, 64 * 8 * 4));
            hipLaunchKernelGGL(t1_load, dim3(1), dim3(64), 0, 0, d, n, shift, o);
            CK(hipDeviceSynchronize());
            std::vector<float> r(64 * 8);
            CK(hipMemcpy(r.data(), o, r.size() * 4, hipMemcpyDeviceToHost));
            int bad = 0;
            for (int l = 0; l < 64; ++l)
                for (int k = 0; k < 8; ++k) {
                    const uint32_t at = shift + 24 * l + (k < 4 ? k : 20 + k - 4);
                    if (r[8 * l + k] != (float)h[at]) ++bad;
                }
            printf("T1 typed load 8_8_8_8 USCALED, byte shift %u: %d of 512 values wrong (lane 1: %g %g %g %g | %g %g %g %g; bytes %u %u %u %u)\n", shift, bad,
                   r[8], r[9], r[10], r[11], r[12], r[13], r[14], r[15], h[shift + 24], h[shift + 25], h[shift + 26], h[shift + 27]);
        }
        hipFree(d); hipFree(o);
    }
    // ---- T2 ----
    {
        const float vals[] = {-1.5f, -0.5f, -0.0f, 0.0f, 0.25f, 0.5f, 0.99f, 1.0f, 1.5f, 2.5f, 3.5f, 127.5f, 254.5f, 254.99f, 255.0f, 255.5f,
                              256.0f, 300.0f, 1e9f, -1e9f, NAN, INFINITY, -INFINITY, 100.75f, -2.5f, -1.49f, -32768.7f, 32767.6f, 32767.4f, -32768.4f, 40000.0f, -40000.0f};
        const int nv = sizeof(vals) / sizeof(float);   // 32 -> 8 lanes
        float *dv; uint8_t *dd;
        CK(hipMalloc(&dv, 64 * 4 * 4)); CK(hipMalloc(&dd, 4096));
        std::vector<float> hv(64 * 4, 0.f);
        std::memcpy(hv.data(), vals, sizeof(vals));
        CK(hipMemcpy(dv, hv.data(), hv.size() * 4, hipMemcpyHostToDevice));
        const char *names[4] = {"8_8_8_8 USCALED", "8_8_8_8 UINT", "16x4 SSCALED", "16x4 SINT"};
        for (int f = 0; f < 4; ++f) {
            CK(hipMemset(dd, 0xee, 4096));
            if (f == 0) hipLaunchKernelGGL(t2_store<0>, dim3(1), dim3(64), 0, 0, dd, 4096u, dv);
            if (f == 1) hipLaunchKernelGGL(t2_store<1>, dim3(1), dim3(64), 0, 0, dd, 4096u, dv);
            if (f == 2) hipLaunchKernelGGL(t2_store<2>, dim3(1), dim3(64), 0, 0, dd, 4096u, dv);
            if (f == 3) hipLaunchKernelGGL(t2_store<3>, dim3(1), dim3(64), 0, 0, dd, 4096u, dv);
            CK(hipDeviceSynchronize());
            std::vector<uint8_t> r(4096);
            CK(hipMemcpy(r.data(), dd, 4096, hipMemcpyDeviceToHost));
            printf("T2 typed store %-16s:", names[f]);
            for (int i = 0; i < nv; ++i) {
                if (f < 2) printf(" %g->%u", vals[i], r[i]);
                else { int16_t s; std::memcpy(&s, &r[2 * i], 2); printf(" %g->%d", vals[i], (int)s); }
            }
            printf("\n");
        }
        hipFree(dv); hipFree(dd);
    }
    // ---- T3 / T4 ----
    {
        const int n = 1 << 20;
        std::vector<float> lo(n), hi(n);
        uint32_t s = 12345;
        auto rnd = [&]() { s = s * 1664525u + 1013904223u; return s; };
        for (int i = 0; i < n; ++i) {
            lo[i] = ((int)(rnd() >> 8) - (1 << 23)) / 256.0f;    // +-32768 with fractions
            hi[i] = ((int)(rnd() >> 8) - (1 << 23)) / 256.0f;
            if (i < 8) { lo[i] = i - 3.5f; hi[i] = -(i - 3.5f); }
        }
        lo[8] = -0.0f; lo[9] = 0.0f; lo[10] = -32767.9f; hi[10] = 32767.9f;
        float *dl, *dh; uint32_t *dp, *dc;
        CK(hipMalloc(&dl, n * 4)); CK(hipMalloc(&dh, n * 4)); CK(hipMalloc(&dp, n * 4)); CK(hipMalloc(&dc, n * 4));
        CK(hipMemcpy(dl, lo.data(), n * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dh, hi.data(), n * 4, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(t3_pack, dim3(n / 256), dim3(256), 0, 0, dl, dh, dp, dc, n);
        CK(hipDeviceSynchronize());
        std::vector<uint32_t> p(n), c(n);
        CK(hipMemcpy(p.data(), dp, n * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(c.data(), dc, n * 4, hipMemcpyDeviceToHost));
        int bad = 0, badc = 0;
        for (int i = 0; i < n; ++i) {
            const int32_t a = (int32_t)lo[i], b = (int32_t)hi[i];   // |x| < 2^15 here except the saturation pair
            const uint32_t want = ((uint32_t)a & 0xffffu) | ((uint32_t)b << 16);
            if (i != 10 && p[i] != want) { if (bad < 4) printf("  T3 mismatch at %d: %g %g -> %08x, want %08x\n", i, lo[i], hi[i], p[i], want); ++bad; }
            float cs = copysignf(0.49999997f, lo[i]); uint32_t cw; std::memcpy(&cw, &cs, 4);
            if (c[i] != cw) { if (badc < 4) printf("  T4 mismatch at %d: %g -> %08x, want %08x\n", i, lo[i], c[i], cw); ++badc; }
        }
        printf("T3 SDWA pair pack (cvt_i32 + cvt_i32_sdwa WORD_1 preserve): %d of %d pairs wrong; first words %08x %08x %08x\n", bad, n, p[0], p[1], p[2]);
        printf("T4 copysign by v_bitop3_b32 0xca: %d of %d wrong\n", badc, n);
        hipFree(dl); hipFree(dh); hipFree(dp); hipFree(dc);
    }

    // ---- T6 ----
    {
        std::vector<uint8_t> h(256);
        for (int i = 0; i < 256; ++i) h[i] = (uint8_t)i;
        uint8_t *d; float *o; uint32_t *raw;
        CK(hipMalloc(&d, 256)); CK(hipMalloc(&o, 256 * 4)); CK(hipMalloc(&raw, 128 * 4));
        CK(hipMemcpy(d, h.data(), 256, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(t6_d16, dim3(1), dim3(64), 0, 0, d, 256u, o, raw);
        CK(hipDeviceSynchronize());
        std::vector<float> r(256); std::vector<uint32_t> rw(128);
        CK(hipMemcpy(r.data(), o, 1024, hipMemcpyDeviceToHost)); CK(hipMemcpy(rw.data(), raw, 512, hipMemcpyDeviceToHost));
        int bad = 0;
        for (int i = 0; i < 256; ++i) { volatile float want = 0.2990f * (float)i; if (r[i] != want) { if (bad < 4) printf("  T6 mismatch: byte %d -> %.9g, want %.9g\n", i, r[i], (float)want); ++bad; } }
        printf("T6 D16 typed load + v_fma_mix_f32 (0.2990 * byte): %d of 256 wrong; raw halves of lane 1: %08x %08x (bytes 4..7)\n", bad, rw[2], rw[3]);
    }
    // ---- T5 ----
    {
        const int W = 4096, H = 4096, tiles_x = W / 256, tiles = tiles_x * (H / 64);
        const int RING = 4;
        std::vector<uint8_t *> frames(RING);
        std::vector<uint8_t> h((size_t)W * H * 3);
        uint32_t s = 99;
        for (size_t i = 0; i < h.size(); ++i) { s = s * 1664525u + 1013904223u; h[i] = (uint8_t)(s >> 24); }
        for (int r = 0; r < RING; ++r) { CK(hipMalloc(&frames[r], h.size())); CK(hipMemcpy(frames[r], h.data(), h.size(), hipMemcpyHostToDevice)); }
        float *o1, *o2;
        CK(hipMalloc(&o1, (size_t)tiles * 256 * 4)); CK(hipMalloc(&o2, (size_t)tiles * 256 * 4));
        auto check = [&](const char *name) {
            std::vector<float> a((size_t)tiles * 256), b((size_t)tiles * 256);
            hipMemcpy(a.data(), o1, a.size() * 4, hipMemcpyDeviceToHost); hipMemcpy(b.data(), o2, b.size() * 4, hipMemcpyDeviceToHost);
            size_t bad = 0;
            for (size_t i = 0; i < a.size(); ++i) bad += a[i] != b[i];
            printf("   %s: %zu of %zu block sums differ from the plain fetch\n", name, bad, a.size());
        };
        for (int rep = 0; rep < 2; ++rep) {
            const float ta = time_us([&](int i) { hipLaunchKernelGGL(t5_plain, dim3(tiles), dim3(256), 0, 0, frames[i % RING], W, tiles_x, o1); }, 40);
            const float tb8 = time_us([&](int i) { hipLaunchKernelGGL(t5_typed<8>, dim3(tiles), dim3(256), 0, 0, frames[i % RING], W, H, tiles_x, o2); }, 40);
            if (rep == 0) check("typed, all rows up front");
            const float tb2 = time_us([&](int i) { hipLaunchKernelGGL(t5_typed<2>, dim3(tiles), dim3(256), 0, 0, frames[i % RING], W, H, tiles_x, o2); }, 40);
            if (rep == 0) check("typed, two rows ahead");
            const float tb3 = time_us([&](int i) { hipLaunchKernelGGL(t5_typed<3>, dim3(tiles), dim3(256), 0, 0, frames[i % RING], W, H, tiles_x, o2); }, 40);
            printf("T5 fetch of a 4096 x 4096 RGB8 frame (50.3 MB): plain + 24 cvt/row %.1f us | typed, all up front %.1f us | typed 2 rows ahead %.1f us | typed 3 rows ahead %.1f us\n",
                   ta, tb8, tb2, tb3);
        }
    }
    return 0;
}
