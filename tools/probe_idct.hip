// probe_idct.hip -- issue efficiency of the dequantise + IDCT instruction stream itself
// (no global memory): each work-item transforms the same in-register block `iters` times.
// Variants: table in LDS (product form) vs table baked as a kernel-arg (SGPR) array.
#include "../jpeg_amd/csrc/dct.hpp"
#include <cstdio>
using namespace jpeg_amd;

struct Tab { float q[64]; };

template <int VAR, int LDS_KB>
__global__ __launch_bounds__(256) void k(uint32_t *out, Tab tab, int iters)
{
    __shared__ float sq[64];
    __shared__ float pad[LDS_KB * 256 + 16];
    if (threadIdx.x < 64) sq[threadIdx.x] = tab.q[threadIdx.x];
    if (iters < 0) pad[threadIdx.x] = 1.0f;
    __syncthreads();
    uint32_t w[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) w[i] = threadIdx.x * 2654435761u + i * 40503u;
    uint32_t acc = 0;
    for (int it = 0; it < iters; ++it) {
        float g[64];
        if (VAR == 0) idct_block(w, sq, 128.5f, g);
        else idct_block(w, tab.q, 128.5f, g);
#pragma unroll
        for (int i = 0; i < 64; i += 2) {
            uint32_t p = __builtin_amdgcn_cvt_pk_u8_f32(floorf(g[i]), 0, 0);
            p = __builtin_amdgcn_cvt_pk_u8_f32(floorf(g[i + 1]), 1, p);
            acc += p;
            w[i >> 1] ^= p;       // make the next iteration depend on this one
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

typedef void (*kfn)(uint32_t *, Tab, int);
int main()
{
    uint32_t *d; (void)hipMalloc(&d, 256 * 4096 * 4);
    Tab t; for (int i = 0; i < 64; ++i) t.q[i] = 0.5f + 0.01f * i;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    struct { const char *name; kfn fn; int wgs; } ks[] = {
        {"table in LDS, 3 waves/SIMD", k<0, 48>, 768}, {"table in LDS, 5 waves/SIMD", k<0, 28>, 1280},
        {"table in SGPRs (kernarg), 3 waves/SIMD", k<1, 48>, 768}, {"table in SGPRs (kernarg), 5 waves/SIMD", k<1, 28>, 1280}};
    for (auto &kk : ks) {
        const int iters = 16 * 1024 * 4 / (kk.wgs * 4) * 4;  // ~ 65536 block-waves in total... keep simple: fixed work
        const int per_wave = 65536 / (kk.wgs * 4) + 1;
        float best = 1e9;
        for (int rep = 0; rep < 3; ++rep) {
            (void)hipEventRecord(e0);
            hipLaunchKernelGGL(kk.fn, dim3(kk.wgs), dim3(256), 0, 0, d, t, per_wave);
            (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1);
            if (ms < best) best = ms;
        }
        const double waves = (double)kk.wgs * 4 * per_wave;
        printf("%-42s %8.1f us for %.0f block-waves  -> %.2f G blocks/s, %.0f cycles per block-wave per SIMD @2.1GHz\n", kk.name,
               best * 1e3, waves, waves * 64 / best / 1e6, best * 1e-3 * 2.1e9 / (waves / 1024));
        (void)iters;
    }
    return 0;
}
