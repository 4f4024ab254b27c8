// probe_sdwa.hip -- (1) issue rate of v_min_f32 / v_max_f32; (2) does
// v_cvt_u32_f32_sdwa ... dst_sel:BYTE_n dst_unused:UNUSED_PRESERVE truncate into one byte and keep the others?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void k_sdwa(const float *in, unsigned *out, int n)
{
    int i = threadIdx.x;
    float v = i < n ? in[i] : 0.f;
    unsigned d = 0xAABBCCDDu;
    asm volatile("v_cvt_u32_f32_sdwa %0, %1 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD" : "+v"(d) : "v"(v));
    unsigned e = 0x11223344u;
    asm volatile("v_cvt_u32_f32_sdwa %0, %1 clamp dst_sel:BYTE_2 dst_unused:UNUSED_PRESERVE src0_sel:DWORD" : "+v"(e) : "v"(v));
    if (i < n) { out[2 * i] = d; out[2 * i + 1] = e; }
}

#define CHAIN8(INSTR)                                                                         \
    asm volatile(INSTR(%0) "\n" INSTR(%1) "\n" INSTR(%2) "\n" INSTR(%3) "\n" INSTR(%4) "\n"     \
                 INSTR(%5) "\n" INSTR(%6) "\n" INSTR(%7)                                        \
                 : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) \
                 : "v"(a), "v"(b))
#define I_MIN(r) "v_min_f32 " #r ", " #r ", %8"
#define I_MAX(r) "v_max_f32 " #r ", " #r ", %9"
#define I_CVTSDWA(r) "v_cvt_u32_f32_sdwa " #r ", " #r " dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD"
#define I_TRUNC(r) "v_trunc_f32 " #r ", " #r
#define I_RNDNE(r) "v_rndne_f32 " #r ", " #r
#define I_FMAMIX(r) "v_fma_mix_f32 " #r ", " #r ", %8, %9"
#define KERNEL(NAME, INSTR)                                                         \
    __global__ __launch_bounds__(256) void NAME(float *out, float a, float b, int iters) \
    {                                                                               \
        float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7; \
        for (int i = 0; i < iters; ++i) { CHAIN8(INSTR); CHAIN8(INSTR); }            \
        out[blockIdx.x * 256 + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;  \
    }
KERNEL(k_min, I_MIN) KERNEL(k_max, I_MAX) KERNEL(k_cvtsdwa, I_CVTSDWA) KERNEL(k_trunc, I_TRUNC) KERNEL(k_rndne, I_RNDNE) KERNEL(k_fmamix, I_FMAMIX)
typedef void (*kfn)(float *, float, float, int);

int main()
{
    std::vector<float> v = {0.0f, 0.6f, 0.999f, 1.5f, 2.5f, 3.99999f, 254.999f, 255.0f, 255.9f, 256.0f, 300.0f, 511.7f, 1e9f, -0.4f, -0.9999f, -3.0f, 127.5f};
    float *d_in; unsigned *d_out;
    (void)hipMalloc(&d_in, 1024); (void)hipMalloc(&d_out, 2048);
    (void)hipMemcpy(d_in, v.data(), v.size() * 4, hipMemcpyHostToDevice);
    k_sdwa<<<1, 64>>>(d_in, d_out, (int)v.size());
    std::vector<unsigned> o(2 * v.size());
    (void)hipMemcpy(o.data(), d_out, o.size() * 4, hipMemcpyDeviceToHost);
    for (size_t i = 0; i < v.size(); ++i) printf("  %14.5f -> byte1 of %08x (no clamp) | byte2 of %08x (clamp)\n", v[i], o[2 * i], o[2 * i + 1]);

    float *d; (void)hipMalloc(&d, 256 * 2048 * 4);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    struct { const char *name; kfn fn; } ks[] = {{"v_min_f32", k_min}, {"v_max_f32", k_max}, {"v_cvt_u32_f32_sdwa byte", k_cvtsdwa},
                                                 {"v_trunc_f32", k_trunc}, {"v_rndne_f32", k_rndne}, {"v_fma_mix_f32", k_fmamix}};
    const int iters = 10000, blocks = 2048;
    for (auto &k : ks) {
        float best = 1e9;
        for (int rep = 0; rep < 3; ++rep) {
            (void)hipEventRecord(e0);
            hipLaunchKernelGGL(k.fn, dim3(blocks), dim3(256), 0, 0, d, 1.0f, 1.0000001f, iters);
            (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1);
            if (ms < best) best = ms;
        }
        printf("%-26s %8.3f ms  %6.2f T instr-lanes/s\n", k.name, best, (double)blocks * 256 * iters * 16 / best / 1e9);
    }
    return 0;
}
