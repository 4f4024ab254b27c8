// probe_phase.hip -- does separating reads and writes IN TIME, chip-wide, beat the mixed copy rate?
// Persistent waves copy 1 GiB in chunks of U x 16 bytes per lane.  With a period P > 0 every wave
// aligns itself on the chip-wide 100 MHz real-time counter (s_memrealtime: no communication):
// loads are issued only in the first half of a period, stores only in the second half.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned int u4 __attribute__((ext_vector_type(4)));

__device__ inline void wait_phase(unsigned period_ticks, unsigned split_ticks, bool second)
{
    // bounded: at most a few periods, never a hang
    for (int spin = 0; spin < 4096; ++spin) {
        const unsigned t = (unsigned)__builtin_amdgcn_s_memrealtime() % period_ticks;
        if ((t >= split_ticks) == second) return;
        __builtin_amdgcn_s_sleep(8);
    }
}

template <int U>
__global__ __launch_bounds__(256) void k_phase(const u4 *src, u4 *dst, size_t n, unsigned period_ticks, unsigned split_ticks)
{
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i + (U - 1) * stride < n; i += U * stride) {
        u4 v[U];
        if (period_ticks) wait_phase(period_ticks, split_ticks, false);
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = src[i + u * stride];
        if (period_ticks) {
            __builtin_amdgcn_s_waitcnt(0);   // the data is here before the write window is asked for
            wait_phase(period_ticks, split_ticks, true);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) dst[i + u * stride] = v[u];
    }
}
// each wave owns contiguous 1 KiB x U chunks (like a strip), not a grid-strided line
template <int U>
__global__ __launch_bounds__(256) void k_phase_chunk(const u4 *src, u4 *dst, size_t n, unsigned period_ticks, unsigned split_ticks)
{
    const size_t wave = (blockIdx.x * 256ull + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    const size_t nwaves = (size_t)gridDim.x * 4;
    for (size_t c = wave; (c + 1) * 64 * U <= n; c += nwaves) {
        u4 v[U];
        if (period_ticks) wait_phase(period_ticks, split_ticks, false);
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = src[c * 64 * U + u * 64 + lane];
        if (period_ticks) {
            __builtin_amdgcn_s_waitcnt(0);
            wait_phase(period_ticks, split_ticks, true);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) dst[c * 64 * U + u * 64 + lane] = v[u];
    }
}
template <typename F> float time_it(F &&f)
{
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float best = 1e9f;
    for (int r = 0; r < 5; ++r) { (void)hipEventRecord(e0); f(); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1); float ms; (void)hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms; }
    return best;
}
int main()
{
    const size_t bytes = 1024ull << 20, n = bytes / 16;
    u4 *a, *b; (void)hipMalloc(&a, bytes); (void)hipMalloc(&b, bytes);
    (void)hipMemset(a, 1, bytes);
    for (int per_cu : {2, 4, 8}) {
        const int grid = 256 * per_cu;
        for (unsigned period_us : {0u, 2u, 4u, 8u, 16u, 32u, 64u}) {
            const unsigned p = period_us * 100, s = p / 2;
            float m8 = time_it([&] { hipLaunchKernelGGL((k_phase<8>), dim3(grid), dim3(256), 0, 0, a, b, n, p, s); });
            float m16 = time_it([&] { hipLaunchKernelGGL((k_phase<16>), dim3(grid), dim3(256), 0, 0, a, b, n, p, s); });
            float c8 = time_it([&] { hipLaunchKernelGGL((k_phase_chunk<8>), dim3(grid), dim3(256), 0, 0, a, b, n, p, s); });
            float c16 = time_it([&] { hipLaunchKernelGGL((k_phase_chunk<16>), dim3(grid), dim3(256), 0, 0, a, b, n, p, s); });
            printf("blocks/CU %d period %2u us: strided U=8 %5.0f U=16 %5.0f | chunked U=8 %5.0f U=16 %5.0f GB/s\n", per_cu, period_us,
                   2.0 * bytes / m8 / 1e6, 2.0 * bytes / m16 / 1e6, 2.0 * bytes / c8 / 1e6, 2.0 * bytes / c16 / 1e6);
        }
    }
    float mc = time_it([&] { (void)hipMemcpyAsync(b, a, bytes, hipMemcpyDeviceToDevice, 0); });
    printf("hipMemcpy D2D 1 GiB: %5.0f GB/s\n", 2.0 * bytes / mc / 1e6);
    return 0;
}
