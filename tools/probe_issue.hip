// probe_issue.hip -- VALU issue rate against the number of waves per SIMD (1 .. 8), for the decode kernels' instruction mix:
// how far below the SIMD's rate does a kernel run that can only keep three waves per SIMD resident?
// Streams: 8 independent chains of one instruction; a stream that alternates VALU with SALU; one with an LDS read per 8 VALU.
#include <hip/hip_runtime.h>
#include <cstdio>

#define R8(S) S S S S S S S S
#define CH8(I) I(%0) I(%1) I(%2) I(%3) I(%4) I(%5) I(%6) I(%7)
#define I_ADD(r) "v_add_f32 " #r ", " #r ", %9\n\t"
#define I_FLOOR(r) "v_floor_f32 " #r ", " #r "\n\t"
#define I_ADD_S(r) "v_add_f32 " #r ", " #r ", %9\n\ts_add_u32 %8, %8, 1\n\t"

#define BODY(NAME, STR, EXTRA)                                                                           \
    __global__ __launch_bounds__(256) void NAME(float *out, float a, int iters)                          \
    {                                                                                                    \
        __shared__ float lds[1024];                                                                      \
        lds[threadIdx.x] = a; __syncthreads();                                                           \
        float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7; \
        unsigned s = 0; float l = 0;                                                                     \
        const float *lp = lds + (threadIdx.x & 63);                                                      \
        for (int i = 0; i < iters; ++i) {                                                                \
            asm volatile(STR : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7), "+s"(s) : "v"(a) : "scc"); \
            EXTRA                                                                                        \
        }                                                                                                \
        out[blockIdx.x * 256 + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + l + (float)s;     \
    }
BODY(k_add, R8(CH8(I_ADD)), )
BODY(k_floor, R8(CH8(I_FLOOR)), )
BODY(k_add_salu, R8(CH8(I_ADD_S)), )
BODY(k_add_lds, R8(CH8(I_ADD)), { float t; t = *(volatile float *)lp; l += t; })

typedef void (*kfn)(float *, float, int);
int main()
{
    float *d; (void)hipMalloc(&d, 256 * 2048 * 4);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    struct { const char *name; kfn fn; double valu_per_iter; } ks[] = {
        {"v_add_f32 (8 chains)", k_add, 64}, {"v_floor_f32 (8 chains)", k_floor, 64},
        {"v_add_f32 + s_add_u32 alternating", k_add_salu, 64}, {"64 v_add_f32 + 1 ds_read_b32 waited for", k_add_lds, 64}};
    const int iters = 500;
    printf("%-42s", "waves per SIMD:");
    for (int w = 1; w <= 8; ++w) printf(" %7d", w);
    fflush(stdout);
    printf("   (VALU instructions per SIMD per 1000 cycles at 2.4 GHz; 500 = one per 2 cycles)\n");
    for (auto &k : ks) {
        printf("%-42s", k.name);
        for (int w = 1; w <= 8; ++w) {
            float best = 1e9;
            for (int rep = 0; rep < 3; ++rep) {
                (void)hipEventRecord(e0);
                hipLaunchKernelGGL(k.fn, dim3(256 * w), dim3(256), 0, 0, d, 1.0f, iters);
                (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
                float ms; (void)hipEventElapsedTime(&ms, e0, e1);
                if (ms < best) best = ms;
            }
            // instructions per SIMD = w waves * iters * valu_per_iter; time in 2.4 GHz cycles
            fflush(stdout);
            printf(" %7.0f", 1000.0 * (w * (double)iters * k.valu_per_iter) / (best * 1e-3 * 2.4e9));
        }
        printf("\n");
    }
    return 0;
}
