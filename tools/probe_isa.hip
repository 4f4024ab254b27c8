// probe_isa.hip -- one-off hardware probes behind design decisions in DESIGN.md.
//   1. rounding / saturation behaviour of v_cvt_pk_u8_f32
//   2. throughput of scalar vs packed f32 add/mul (v_pk_add_f32 / v_pk_mul_f32)
// build: hipcc --offload-arch=gfx950 -O3 -o probe_isa tools/probe_isa.hip ; run on the GPU box
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void k_cvt(const float *in, unsigned *out, int n)
{
    int i = threadIdx.x;
    if (i < n) out[i] = __builtin_amdgcn_cvt_pk_u8_f32(in[i], 0u, 0u);
}

typedef float float2_ __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(256) void k_alu(float *out, float a, float b, int iters)
{
    float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    if (MODE == 0) {
        for (int i = 0; i < iters; ++i) {
            asm volatile(
                "v_add_f32 %0, %0, %8\n v_add_f32 %1, %1, %8\n v_add_f32 %2, %2, %8\n v_add_f32 %3, %3, %8\n"
                "v_mul_f32 %4, %4, %9\n v_mul_f32 %5, %5, %9\n v_mul_f32 %6, %6, %9\n v_mul_f32 %7, %7, %9\n"
                : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b));
        }
    } else {
        float2_ p0 = {x0, x1}, p1 = {x2, x3}, p2 = {x4, x5}, p3 = {x6, x7}, pa = {a, a}, pb = {b, b};
        for (int i = 0; i < iters; ++i) {
            asm volatile(
                "v_pk_add_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %4\n v_pk_mul_f32 %2, %2, %5\n v_pk_mul_f32 %3, %3, %5\n"
                "v_pk_add_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %4\n v_pk_mul_f32 %2, %2, %5\n v_pk_mul_f32 %3, %3, %5\n"
                : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pa), "v"(pb));
        }
        x0 = p0.x; x1 = p0.y; x2 = p1.x; x3 = p1.y; x4 = p2.x; x5 = p2.y; x6 = p3.x; x7 = p3.y;
    }
    out[blockIdx.x * 256 + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
}

int main()
{
    std::vector<float> v = {0.0f, 0.4f, 0.5f, 0.6f, 0.999f, 1.5f, 2.5f, 3.5f, 254.5f, 254.999f, 255.0f, 255.4f,
                            255.5f, 256.0f, 300.0f, 1e9f, -0.4f, -0.6f, -3.0f, 127.5f, 128.5f};
    float *d_in; unsigned *d_out;
    hipMalloc(&d_in, 256 * 4); hipMalloc(&d_out, 256 * 4);
    hipMemcpy(d_in, v.data(), v.size() * 4, hipMemcpyHostToDevice);
    k_cvt<<<1, 64>>>(d_in, d_out, (int)v.size());
    std::vector<unsigned> o(v.size());
    hipMemcpy(o.data(), d_out, v.size() * 4, hipMemcpyDeviceToHost);
    printf("v_cvt_pk_u8_f32:\n");
    for (size_t i = 0; i < v.size(); ++i) printf("  %12.4f -> %u\n", v[i], o[i] & 0xff);

    float *d_f; hipMalloc(&d_f, 256 * 2048 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000, blocks = 2048;
    for (int mode = 0; mode < 2; ++mode) {
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            if (mode == 0) k_alu<0><<<blocks, 256>>>(d_f, 1.0f, 1.0000001f, iters);
            else k_alu<1><<<blocks, 256>>>(d_f, 1.0f, 1.0000001f, iters);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            // lane-ops: scalar mode 8 ops/iter/lane; packed mode 8 instr x 2 = 16 lane-ops/iter/lane
            double laneops = (double)blocks * 256 * iters * (mode == 0 ? 8 : 16);
            printf("mode %s rep %d: %.3f ms  %.2f T lane-ops/s (%.2f T instr-lanes/s)\n", mode ? "packed" : "scalar", rep, ms,
                   laneops / ms / 1e9, laneops / ms / 1e9 / (mode ? 2 : 1));
        }
    }
    return 0;
}
