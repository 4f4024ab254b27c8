// probe_launch.hip -- how long does the dispatcher take to get a persistent grid onto the chip?  (round 4)
// 768 workgroups of 256 work-items with 51 KiB of LDS each (k_quad420's launch) against 256 workgroups of 768 work-items with
// 153 KiB: every wave records the chip-wide 100 MHz counter when it starts; reported: mean / max start of a wave relative to
// the first one, per launch shape.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>

template <int THREADS, int LDS_DWORDS>
__global__ __launch_bounds__(THREADS) void k_start(unsigned long long *t, float *sink)
{
    __shared__ float buf[LDS_DWORDS];
    const unsigned long long now = __builtin_amdgcn_s_memrealtime();
    buf[threadIdx.x] = (float)now;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) t[(blockIdx.x * THREADS + threadIdx.x) >> 6] = now;
    // stay resident for a while so that the whole grid has to be co-resident
    float x = buf[(threadIdx.x * 7) % LDS_DWORDS];
    for (int i = 0; i < 20000; ++i) x = x * 1.0000001f + 0.5f;
    if (x == 12345.0f) sink[0] = x;
}

template <int THREADS, int LDS_DWORDS>
void run(const char *name, int grid, unsigned long long *d_t, float *d_sink)
{
    const int waves = grid * THREADS / 64;
    std::vector<unsigned long long> h(waves);
    double mean_acc = 0, max_acc = 0;
    const int reps = 20;
    for (int r = 0; r < reps + 2; ++r) {
        hipLaunchKernelGGL((k_start<THREADS, LDS_DWORDS>), dim3(grid), dim3(THREADS), 0, 0, d_t, d_sink);
        (void)hipDeviceSynchronize();
        (void)hipMemcpy(h.data(), d_t, waves * 8, hipMemcpyDeviceToHost);
        const unsigned long long t0 = *std::min_element(h.begin(), h.end());
        double m = 0, mx = 0;
        for (auto v : h) { m += (double)(v - t0); mx = std::max(mx, (double)(v - t0)); }
        if (r >= 2) { mean_acc += m / waves / 100.0; max_acc += mx / 100.0; }
    }
    printf("%-44s grid %4d x %4d: start of a wave mean %5.2f us, last %5.2f us (mean of %d launches)\n", name, grid, THREADS, mean_acc / reps, max_acc / reps, reps);
}

// the same grid as TWO launches on two streams (half the workgroups each), issued back to back: does a second hardware queue
// halve the ramp?
template <int THREADS, int LDS_DWORDS>
void run_two_streams(const char *name, int grid, unsigned long long *d_t, float *d_sink)
{
    hipStream_t s1, s2; (void)hipStreamCreateWithFlags(&s1, hipStreamNonBlocking); (void)hipStreamCreateWithFlags(&s2, hipStreamNonBlocking);
    const int waves = grid * THREADS / 64, half = grid / 2;
    std::vector<unsigned long long> h(waves);
    double mean_acc = 0, max_acc = 0;
    const int reps = 20;
    for (int r = 0; r < reps + 2; ++r) {
        hipLaunchKernelGGL((k_start<THREADS, LDS_DWORDS>), dim3(half), dim3(THREADS), 0, s1, d_t, d_sink);
        hipLaunchKernelGGL((k_start<THREADS, LDS_DWORDS>), dim3(grid - half), dim3(THREADS), 0, s2, d_t + (size_t)half * THREADS / 64, d_sink);
        (void)hipDeviceSynchronize();
        (void)hipMemcpy(h.data(), d_t, waves * 8, hipMemcpyDeviceToHost);
        const unsigned long long t0 = *std::min_element(h.begin(), h.end());
        double m = 0, mx = 0;
        for (auto v : h) { m += (double)(v - t0); mx = std::max(mx, (double)(v - t0)); }
        if (r >= 2) { mean_acc += m / waves / 100.0; max_acc += mx / 100.0; }
    }
    printf("%-44s grid 2 x %4d x %4d: start of a wave mean %5.2f us, last %5.2f us (mean of %d launches)\n", name, half, THREADS, mean_acc / reps, max_acc / reps, reps);
}

int main()
{
    unsigned long long *d_t; float *d_sink;
    (void)hipMalloc(&d_t, 8 * 8192); (void)hipMalloc(&d_sink, 4);
    run<256, 12800>("3 workgroups per CU (256 threads, 51 KB LDS)", 768, d_t, d_sink);
    run<768, 38400>("1 workgroup per CU (768 threads, 153 KB LDS)", 256, d_t, d_sink);
    run<512, 19200>("2 workgroups per CU (512 threads, 77 KB LDS)", 512, d_t, d_sink);
    run<256, 256>("3 per CU, no LDS to speak of", 768, d_t, d_sink);
    run<1024, 38400>("1 per CU, 1024 threads, 153 KB", 256, d_t, d_sink);
    run_two_streams<256, 12800>("3 per CU as two launches on two streams", 768, d_t, d_sink);
    run<256, 6400>("4 per CU (256 threads, 25 KB: the encode)", 1024, d_t, d_sink);
    run_two_streams<256, 6400>("4 per CU as two launches on two streams", 1024, d_t, d_sink);
    return 0;
}
