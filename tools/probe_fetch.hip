// probe_fetch.hip -- is straight-line code (K2 is ~20 KB of instructions per strip, fully
// unrolled) slower to issue than the same instruction mix in a tight loop?  Same 22-instruction
// block as probe_mix.hip, either looped 123x (176 B of code) or unrolled 123x (~21 KB) inside
// an outer loop of 4 "strips"; 3 waves/SIMD resident.  Also: the unrolled block with every
// instruction forced into an 8-byte encoding (VOP3), to see fetch-bandwidth sensitivity.
#include <hip/hip_runtime.h>
#include <cstdio>

#define BLOCK22 \
    "v_add_f32 %0, %0, %10\n v_mul_f32 %1, %1, %11\n v_sub_f32 %2, %2, %10\n v_fma_f32 %3, %3, %11, %10\n" \
    "v_floor_f32 %4, %0\n v_cvt_pk_u8_f32 %8, %4, 0, %8\n" \
    "v_add_f32 %5, %5, %10\n v_mul_f32 %6, %6, %11\n v_sub_f32 %7, %7, %10\n v_fma_f32 %0, %0, %11, %10\n" \
    "v_floor_f32 %4, %1\n v_cvt_pk_u8_f32 %8, %4, 1, %8\n v_med3_f32 %5, %5, %10, %11\n" \
    "v_add_f32 %1, %1, %10\n v_mul_f32 %2, %2, %11\n v_sub_f32 %3, %3, %10\n v_fma_f32 %6, %6, %11, %10\n" \
    "v_floor_f32 %4, %2\n v_cvt_pk_u8_f32 %9, %4, 2, %9\n v_cvt_f32_ubyte0 %7, %8\n" \
    "v_floor_f32 %4, %3\n v_cvt_pk_u8_f32 %9, %4, 3, %9\n"
#define BLOCK22_E64 \
    "v_add_f32_e64 %0, %0, %10\n v_mul_f32_e64 %1, %1, %11\n v_sub_f32_e64 %2, %2, %10\n v_fma_f32 %3, %3, %11, %10\n" \
    "v_floor_f32_e64 %4, %0\n v_cvt_pk_u8_f32 %8, %4, 0, %8\n" \
    "v_add_f32_e64 %5, %5, %10\n v_mul_f32_e64 %6, %6, %11\n v_sub_f32_e64 %7, %7, %10\n v_fma_f32 %0, %0, %11, %10\n" \
    "v_floor_f32_e64 %4, %1\n v_cvt_pk_u8_f32 %8, %4, 1, %8\n v_med3_f32 %5, %5, %10, %11\n" \
    "v_add_f32_e64 %1, %1, %10\n v_mul_f32_e64 %2, %2, %11\n v_sub_f32_e64 %3, %3, %10\n v_fma_f32 %6, %6, %11, %10\n" \
    "v_floor_f32_e64 %4, %2\n v_cvt_pk_u8_f32 %9, %4, 2, %9\n v_cvt_f32_ubyte0_e64 %7, %8\n" \
    "v_floor_f32_e64 %4, %3\n v_cvt_pk_u8_f32 %9, %4, 3, %9\n"
#define R3(x) x x x
#define R41(x) R3(R3(R3(x))) R3(R3(x)) R3(x) x x   /* 27 + 9 + 3 + 2 = 41 */
#define R123(x) R3(R41(x))
#define OPS : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7), "+v"(p0), "+v"(p1) : "v"(a), "v"(b)

template <int MODE>
__global__ __launch_bounds__(256) void k(float *out, float a, float b, int strips)
{
    __shared__ float pad[48 * 256 + 64];   // 48 KB per workgroup -> 3 workgroups / CU -> 3 waves / SIMD
    float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    unsigned p0 = 0, p1 = 0;
    if (strips < 0) pad[threadIdx.x] = a;
    for (int s = 0; s < strips; ++s) {
        if (MODE == 0) { for (int i = 0; i < 123; ++i) asm volatile(BLOCK22 OPS); }
        else if (MODE == 1) { asm volatile(R123(BLOCK22) OPS); }
        else { asm volatile(R123(BLOCK22_E64) OPS); }
    }
    out[blockIdx.x * 256 + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + p0 + p1;
}
typedef void (*kfn)(float *, float, float, int);
int main()
{
    float *d; (void)hipMalloc(&d, 256 * 4096 * 4);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    struct { const char *name; kfn fn; } ks[] = {{"looped (176 B of code)", k<0>}, {"unrolled (~17 KB of code)", k<1>}, {"unrolled, all 8-byte encodings (~22 KB)", k<2>}};
    // persistent-style: 768 workgroups x 4 waves, each wave 5.33 strips -> use 4096 WG x 1 strip-equivalent: here 768 WGs x 5 strips + remainder
    for (auto &kk : ks) {
        float best = 1e9;
        for (int rep = 0; rep < 3; ++rep) {
            (void)hipEventRecord(e0);
            hipLaunchKernelGGL(kk.fn, dim3(768), dim3(256), 0, 0, d, 1.0f, 1.0000001f, 16 / 3 + 1);  // 6 strips per wave: 768*4*6 = 18432 strip-waves
            (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1);
            if (ms < best) best = ms;
        }
        printf("%-44s %8.1f us for 18432 strip-waves of 2706 VALU (%.2f cycles/instr/SIMD at 2.1 GHz)\n", kk.name, best * 1e3,
               best * 1e-3 * 2.1e9 / (18.0 * 2706));
    }
    return 0;
}
