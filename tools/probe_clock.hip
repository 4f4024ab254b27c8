// probe_clock.hip -- does the chip hold its shader clock under the decode's kind of load?
// Each kernel runs long enough (~ms) for the power management to settle and reports the EFFECTIVE shader
// clock of that run: shader cycles (s_memtime) per 100 MHz reference tick (s_memrealtime), measured by
// one wave per workgroup over the whole kernel.
//   valu    K2's VALU mix (55 % fast f32, 45 % cvt / floor / med3 / pack), no memory traffic
//   stream  read 16 B / write 16 B per lane-iteration from / to HBM (nt stores), no arithmetic
//   both    the VALU mix AND the stream in the same waves, ~ the decode's ratio of 100 VALU per 16 B in + 16 B out
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned int u4 __attribute__((ext_vector_type(4)));

struct Stamp { unsigned long long c0, c1, r0, r1; };

#define MIX(x0, x1, x2, x3, x4, x5, x6, x7, p0, p1, a, b)                                                     \
    asm volatile(                                                                                             \
        "v_add_f32 %0, %0, %10\n v_mul_f32 %1, %1, %11\n v_sub_f32 %2, %2, %10\n v_fma_f32 %3, %3, %11, %10\n" \
        "v_floor_f32 %4, %0\n v_cvt_pk_u8_f32 %8, %4, 0, %8\n"                                                 \
        "v_add_f32 %5, %5, %10\n v_mul_f32 %6, %6, %11\n v_sub_f32 %7, %7, %10\n v_fma_f32 %0, %0, %11, %10\n" \
        "v_floor_f32 %4, %1\n v_cvt_pk_u8_f32 %8, %4, 1, %8\n v_med3_f32 %5, %5, %10, %11\n"                    \
        "v_add_f32 %1, %1, %10\n v_mul_f32 %2, %2, %11\n v_sub_f32 %3, %3, %10\n v_fma_f32 %6, %6, %11, %10\n" \
        "v_floor_f32 %4, %2\n v_cvt_pk_u8_f32 %9, %4, 2, %9\n v_cvt_f32_ubyte0 %7, %8\n"                        \
        "v_floor_f32 %4, %3\n v_cvt_pk_u8_f32 %9, %4, 3, %9\n"                                                 \
        : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7), "+v"(p0), "+v"(p1)  \
        : "v"(a), "v"(b))

// MODE 1 valu, 2 stream, 3 both.  ITERS_V: MIX blocks (22 VALU) per memory iteration.
template <int MODE, int ITERS_V>
__global__ __launch_bounds__(256) void k(const u4 *src, u4 *dst, size_t n, int outer, float a, float b, Stamp *st, float *sink)
{
    __shared__ float pad[12 * 1024];   // 48 KiB: three workgroups per CU, like k_luma_fused
    if (outer < 0) pad[threadIdx.x] = a;
    unsigned long long c0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
    float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    unsigned p0 = 0, p1 = 0;
    const size_t stride = (size_t)gridDim.x * 256;
    size_t i = blockIdx.x * 256ull + threadIdx.x;
    for (int o = 0; o < outer; ++o) {
        u4 v = {0, 0, 0, 0};
        if (MODE & 2) { v = src[i]; }
        if (MODE & 1) {
#pragma unroll 1
            for (int j = 0; j < ITERS_V; ++j) MIX(x0, x1, x2, x3, x4, x5, x6, x7, p0, p1, a, b);
        }
        if (MODE & 2) {
            v.x ^= p0;
            __builtin_nontemporal_store(v, dst + i);
            i += stride; if (i >= n) i -= n;
        }
    }
    unsigned long long c1 = __builtin_readcyclecounter(), r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) st[blockIdx.x] = Stamp{c0, c1, r0, r1};
    if (x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + p0 + p1 == 12345.678f) sink[0] = x0;
}

typedef void (*kfn)(const u4 *, u4 *, size_t, int, float, float, Stamp *, float *);

int main()
{
    const size_t bytes = 1024ull << 20, n = bytes / 16;
    u4 *src, *dst; float *sink; Stamp *d_st;
    (void)hipMalloc(&src, bytes); (void)hipMalloc(&dst, bytes); (void)hipMalloc(&sink, 64); (void)hipMalloc(&d_st, 768 * sizeof(Stamp));
    (void)hipMemset(src, 0x5a, bytes);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    struct { const char *name; kfn fn; int outer; double valu_per_outer; bool mem; } ks[] = {
        {"valu only (K2 mix)", k<1, 5>, 6000, 5 * 22.0, false},
        {"stream only (16 B in + 16 B out)", k<2, 5>, 6000, 0, true},
        {"both, 110 VALU per 32 B", k<3, 5>, 6000, 5 * 22.0, true},
        {"both, 220 VALU per 32 B", k<3, 10>, 3000, 10 * 22.0, true},
        {"both, 66 VALU per 32 B", k<3, 3>, 6000, 3 * 22.0, true},
    };
    for (auto &kk : ks) {
        for (int rep = 0; rep < 2; ++rep) {
            (void)hipEventRecord(e0);
            hipLaunchKernelGGL(kk.fn, dim3(768), dim3(256), 0, 0, src, dst, n, kk.outer, 1.0f, 1.0000001f, d_st, sink);
            (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1);
            if (rep == 0) continue;
            std::vector<Stamp> st(768);
            (void)hipMemcpy(st.data(), d_st, 768 * sizeof(Stamp), hipMemcpyDeviceToHost);
            double ratio = 0;
            for (auto &s : st) ratio += (double)(s.c1 - s.c0) / (double)(s.r1 - s.r0);
            ratio /= 768;
            const double instr = 768.0 * 4 * kk.outer * kk.valu_per_outer;
            const double gb = kk.mem ? 768.0 * 256 * kk.outer * 32 / 1e9 : 0;
            printf("%-36s %8.2f ms  shader clock %5.0f MHz  %6.2f ns per VALU wave-instr per SIMD  %6.0f GB/s\n", kk.name, ms, ratio * 100.0,
                   instr > 0 ? ms * 1e6 / (instr / 1024) : 0.0, gb / (ms * 1e-3));
        }
    }
    return 0;
}
