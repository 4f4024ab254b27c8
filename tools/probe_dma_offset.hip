// probe_dma_offset.hip -- global_load_lds_dwordx4 with an instruction offset: does the offset move the LDS destination as well
// as the global source?  (If it does, the 1 KiB pieces of a contiguous run need ONE M0 write and ONE base address.)
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ void k(const unsigned *g, unsigned *out)
{
    __shared__ __attribute__((aligned(16))) unsigned lds[2048];   // 8 KiB
    for (int i = threadIdx.x; i < 2048; i += 64) lds[i] = 0xdeadbeefu;
    __syncthreads();
    const unsigned base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned *)lds;
    const unsigned sbase = __builtin_amdgcn_readfirstlane(base);
    const unsigned voff = threadIdx.x * 16;
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\t"
                 "global_load_lds_dwordx4 %1, %0\n\t"
                 "global_load_lds_dwordx4 %1, %0 offset:1024\n\t"
                 "global_load_lds_dwordx4 %1, %0 offset:3072\n\t"
                 "s_waitcnt vmcnt(0)" ::"s"(g), "v"(voff), "s"(sbase) : "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 2048; i += 64) out[i] = lds[i];
}

int main()
{
    unsigned *g, *o, h[2048];
    (void)hipMalloc(&g, 8192 * 4); (void)hipMalloc(&o, 2048 * 4);
    unsigned src[8192];
    for (int i = 0; i < 8192; ++i) src[i] = i;   // dword index
    (void)hipMemcpy(g, src, sizeof(src), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, g, o);
    (void)hipMemcpy(h, o, sizeof(h), hipMemcpyDeviceToHost);
    for (int kb = 0; kb < 8; ++kb) {
        printf("LDS KiB %d: ", kb);
        if (h[256 * kb] == 0xdeadbeefu) printf("untouched\n");
        else printf("global dwords %u .. %u (= global byte offset %u)\n", h[256 * kb], h[256 * kb + 255], 4 * h[256 * kb]);
    }
    return 0;
}
