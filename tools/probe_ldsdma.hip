// probe_ldsdma.hip -- what does issuing a burst of global_load_lds_dwordx4 cost the issuing wave, and does anything but
// vmcnt wait for the data?  One wave per SIMD (256 blocks x 256 threads) or 3 (768 blocks); each wave: t0, 8 DMA
// instructions (1 KiB each, its own 8 KiB of LDS), t1 after s_waitcnt lgkmcnt(0), t2 after one ds_read_b32 of an
// UNRELATED LDS word + lgkmcnt(0), t3 after s_waitcnt vmcnt(0).  Prints mean cycles of each interval.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__device__ __forceinline__ void dma16(uint64_t sbase, uint32_t voff, uint32_t lds)
{
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %0 nt" ::"s"(sbase), "v"(voff), "s"(lds) : "memory");
}

template <int MODE>
__global__ __launch_bounds__(256) void k(const char *src, unsigned long long *out, int reps, size_t span)
{
    __shared__ __attribute__((aligned(16))) uint32_t buf[4][2048];
    __shared__ uint32_t other[256];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    other[threadIdx.x] = threadIdx.x;
    __syncthreads();
    const uint32_t lds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint32_t *)buf[wave];
    unsigned long long acc[3] = {0, 0, 0};
    uint32_t sink = 0;
    for (int r = 0; r < reps; ++r) {
        const size_t off = ((size_t)(blockIdx.x * 4 + wave) * reps + r) * 8192 % span;
        const uint64_t sb = __builtin_amdgcn_readfirstlane((uint32_t)((uint64_t)(src + off))) | ((uint64_t)__builtin_amdgcn_readfirstlane((uint32_t)((uint64_t)(src + off) >> 32)) << 32);
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        const unsigned long long t0 = __builtin_readcyclecounter();
#pragma unroll
        for (int i = 0; i < 8; ++i) dma16(sb, 16u * lane + 1024u * i, __builtin_amdgcn_readfirstlane(lds + 1024 * i));
        if (MODE == 1) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        const unsigned long long t1 = __builtin_readcyclecounter();
        uint32_t v;
        asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"((uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint32_t *)&other[threadIdx.x]) : "memory");
        const unsigned long long t2 = __builtin_readcyclecounter();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned long long t3 = __builtin_readcyclecounter();
        acc[0] += t1 - t0; acc[1] += t2 - t1; acc[2] += t3 - t2;
        sink += v + buf[wave][lane];
    }
    if (lane == 0) {
        unsigned long long *o = out + (size_t)(blockIdx.x * 4 + wave) * 4;
        o[0] = acc[0]; o[1] = acc[1]; o[2] = acc[2]; o[3] = sink;
    }
}

int main()
{
    const size_t span = (size_t)1 << 30;
    char *src; (void)hipMalloc(&src, span); (void)hipMemset(src, 1, span);
    unsigned long long *out; (void)hipMalloc(&out, 4096 * 4 * 8);
    const int reps = 200;
    for (int mode = 0; mode < 2; ++mode)
        for (int blocks : {256, 768}) {
            if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, src, out, reps, span);
            else hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, src, out, reps, span);
            (void)hipDeviceSynchronize();
            std::vector<unsigned long long> h(blocks * 16);
            (void)hipMemcpy(h.data(), out, h.size() * 8, hipMemcpyDeviceToHost);
            double a[3] = {0, 0, 0};
            for (int w = 0; w < blocks * 4; ++w) for (int j = 0; j < 3; ++j) a[j] += (double)h[w * 4 + j];
            for (double &x : a) x /= (double)blocks * 4 * reps;
            printf("%d wave(s)/SIMD, %s: 8 DMA issue%s %7.0f cycles | unrelated ds_read + lgkmcnt(0) %7.0f | then vmcnt(0) %7.0f\n", blocks / 256,
                   mode ? "lgkmcnt(0) after the burst" : "no wait after the burst  ", mode ? " + lgkmcnt(0)" : "             ", a[0], a[1], a[2]);
        }
    return 0;
}
