// probe_mix.hip -- how fast can a SIMD issue K2's VALU mix (55% "fast" f32 add/mul/fma,
// 45% "slow" floor/cvt/med3/pack) at a given occupancy?  No memory traffic at all.
#include <hip/hip_runtime.h>
#include <cstdio>

template <int LDS_KB>
__global__ __launch_bounds__(256) void k_mix(float *out, float a, float b, int iters)
{
    __shared__ float pad[LDS_KB * 256 + 64];
    float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    unsigned p0 = 0, p1 = 0;
    if (iters < 0) pad[threadIdx.x] = a;   // keep the allocation
    for (int i = 0; i < iters; ++i) {
        // 12 fast + 10 slow per iteration
        asm volatile(
            "v_add_f32 %0, %0, %10\n v_mul_f32 %1, %1, %11\n v_sub_f32 %2, %2, %10\n v_fma_f32 %3, %3, %11, %10\n"
            "v_floor_f32 %4, %0\n v_cvt_pk_u8_f32 %8, %4, 0, %8\n"
            "v_add_f32 %5, %5, %10\n v_mul_f32 %6, %6, %11\n v_sub_f32 %7, %7, %10\n v_fma_f32 %0, %0, %11, %10\n"
            "v_floor_f32 %4, %1\n v_cvt_pk_u8_f32 %8, %4, 1, %8\n v_med3_f32 %5, %5, %10, %11\n"
            "v_add_f32 %1, %1, %10\n v_mul_f32 %2, %2, %11\n v_sub_f32 %3, %3, %10\n v_fma_f32 %6, %6, %11, %10\n"
            "v_floor_f32 %4, %2\n v_cvt_pk_u8_f32 %9, %4, 2, %9\n v_cvt_f32_ubyte0 %7, %8\n"
            "v_floor_f32 %4, %3\n v_cvt_pk_u8_f32 %9, %4, 3, %9\n"
            : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7), "+v"(p0), "+v"(p1)
            : "v"(a), "v"(b));
    }
    out[blockIdx.x * 256 + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + p0 + p1;
}

typedef void (*kfn)(float *, float, float, int);
int main()
{
    float *d; (void)hipMalloc(&d, 256 * 4096 * 4);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    struct { const char *name; kfn fn; } ks[] = {
        {"occupancy 8 waves/SIMD (no LDS)", k_mix<0>}, {"4 waves/SIMD (36 KB LDS/WG)", k_mix<36>},
        {"3 waves/SIMD (48 KB)", k_mix<48>}, {"2 waves/SIMD (64 KB)", k_mix<64>}, {"1 wave/SIMD (100 KB)", k_mix<100>}};
    // K2-sized: 4096 WGs x 4 waves, each wave 2712 VALU = 123 iterations of 22
    const int iters = 123, blocks = 4096;
    for (auto &k : ks) {
        float best = 1e9;
        for (int rep = 0; rep < 3; ++rep) {
            (void)hipEventRecord(e0);
            hipLaunchKernelGGL(k.fn, dim3(blocks), dim3(256), 0, 0, d, 1.0f, 1.0000001f, iters);
            (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1);
            if (ms < best) best = ms;
        }
        printf("%-36s %8.1f us for %d wave-instr per wave (%.2f cycles/instr/SIMD at 2.1 GHz)\n", k.name, best * 1e3, iters * 22,
               best * 1e-3 * 2.1e9 / (16.0 * iters * 22));
    }
    return 0;
}
