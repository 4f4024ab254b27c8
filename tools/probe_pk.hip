// probe_pk.hip -- issue rate of the packed-FP32 VALU instructions (two binary32 per lane per
// instruction) against their scalar forms.  8 independent chains of ONE instruction per lane.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));

#define CH8(INSTR)                                                                              \
    asm volatile(INSTR(%0) "\n" INSTR(%1) "\n" INSTR(%2) "\n" INSTR(%3) "\n" INSTR(%4) "\n"       \
                 INSTR(%5) "\n" INSTR(%6) "\n" INSTR(%7)                                          \
                 : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) \
                 : "v"(a), "v"(b))
#define I_ADD(r) "v_add_f32 " #r ", " #r ", %8"
#define I_PKADD(r) "v_pk_add_f32 " #r ", " #r ", %8"
#define I_PKMUL(r) "v_pk_mul_f32 " #r ", " #r ", %9"
#define I_PKFMA(r) "v_pk_fma_f32 " #r ", " #r ", %9, %8"

#define KERNEL(NAME, T, INSTR)                                                                   \
    __global__ __launch_bounds__(256) void NAME(float *out, float fa, float fb, int iters)        \
    {                                                                                             \
        T a = (T)fa, b = (T)fb;                                                                   \
        T x0 = (T)(float)threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7; \
        for (int i = 0; i < iters; ++i) { CH8(INSTR); CH8(INSTR); }                                \
        T s = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;                                              \
        out[blockIdx.x * 256 + threadIdx.x] = sum(s);                                             \
    }
__device__ inline float sum(float v) { return v; }
__device__ inline float sum(f2 v) { return v.x + v.y; }
KERNEL(k_add, float, I_ADD) KERNEL(k_pkadd, f2, I_PKADD) KERNEL(k_pkmul, f2, I_PKMUL) KERNEL(k_pkfma, f2, I_PKFMA)
typedef void (*kfn)(float *, float, float, int);
int main()
{
    float *d; (void)hipMalloc(&d, 256 * 4096 * 4);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    struct { const char *name; kfn fn; int width; } ks[] = {
        {"v_add_f32", k_add, 1}, {"v_pk_add_f32", k_pkadd, 2}, {"v_pk_mul_f32", k_pkmul, 2}, {"v_pk_fma_f32", k_pkfma, 2}};
    const int iters = 10000;
    for (int blocks : {256 * 1, 256 * 2, 256 * 3, 256 * 4, 256 * 8}) {   // 1, 2, 3, 4, 8 waves per SIMD
        for (auto &k : ks) {
            float best = 1e9;
            for (int rep = 0; rep < 3; ++rep) {
                (void)hipEventRecord(e0);
                hipLaunchKernelGGL(k.fn, dim3(blocks), dim3(256), 0, 0, d, 1.0f, 1.0000001f, iters);
                (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
                float ms; (void)hipEventElapsedTime(&ms, e0, e1);
                if (ms < best) best = ms;
            }
            const double instr = (double)blocks * 256 * iters * 16;
            printf("%d waves/SIMD  %-14s %8.3f ms  %6.2f T instr-lanes/s  %6.2f T element-ops/s\n", blocks / 256, k.name, best,
                   instr / best / 1e9, instr * k.width / best / 1e9);
        }
    }
    return 0;
}
