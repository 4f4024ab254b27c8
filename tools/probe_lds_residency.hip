// probe_lds_residency.hip -- how many workgroups of 256 work-items with X bytes of LDS are resident on the chip AT ONCE?
// Every workgroup notes the tick (100 MHz) at which it starts, spins ~40 us, and the host counts the workgroups whose start lies within 10 us
// of the first one: the resident capacity as the hardware sees it (hipOccupancyMaxActiveBlocksPerMultiprocessor is what the runtime computes).
//   hipcc --offload-arch=gfx950 -O3 -o tools/probe_lds_residency tools/probe_lds_residency.hip && tools/probe_lds_residency
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>

__global__ __launch_bounds__(256) void k_hold(unsigned long long *start, int spin_ticks)
{
    extern __shared__ unsigned int lds[];
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) start[blockIdx.x] = t0;
    lds[threadIdx.x] = threadIdx.x;
    while (__builtin_amdgcn_s_memrealtime() - t0 < (unsigned long long)spin_ticks) __builtin_amdgcn_s_sleep(8);
    if (lds[(threadIdx.x + 1) & 255] == 0xffffffffu) start[blockIdx.x] = 0;
}

int main()
{
    int cus = 0;
    hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    const int n = 8 * cus;
    unsigned long long *d;
    hipMalloc(&d, n * sizeof(unsigned long long));
    std::vector<unsigned long long> h(n);
    printf("# %d CUs; %d workgroups of 256 work-items launched per size, each holds its CU for 40 us\n# LDS bytes   runtime says   started within 10 us (per CU)\n", cus, n);
    const int sizes[] = {16384, 24576, 32768, 36864, 38912, 40320, 40960, 41984, 46464, 49152, 52224, 53248, 54272, 57344, 65536, 81920};
    for (int bytes : sizes) {
        hipFuncSetAttribute(reinterpret_cast<const void *>(k_hold), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
        int occ = 0;
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k_hold, 256, bytes);
        hipMemset(d, 0, n * sizeof(unsigned long long));
        hipLaunchKernelGGL(k_hold, dim3(n), dim3(256), bytes, 0, d, 4000);
        hipDeviceSynchronize();
        hipMemcpy(h.data(), d, n * sizeof(unsigned long long), hipMemcpyDeviceToHost);
        const unsigned long long first = *std::min_element(h.begin(), h.end());
        int early = 0;
        for (auto t : h) early += t - first < 1000;
        printf("%9d   %2d per CU     %5d  (%.2f per CU)\n", bytes, occ, early, (double)early / cus);
    }
    return 0;
}
