// probe_buffer.hip -- round 5: what buffer (SRD) addressing does on gfx950, before k_quad420 / k_encode_fused rely on it.
//   T1  range check of buffer_store_dwordx4: is `soffset` part of the checked offset?  are out-of-range lanes dropped?
//   T2  buffer_load_dwordx4 ... lds: LDS destination = M0 + inst_offset + 16 lane?  source = base + soffset + voffset + inst_offset?
//       what do out-of-range lanes leave in LDS (zeros / untouched)?
//   T3  unaligned global / buffer loads and stores of 4, 8, 16 bytes (address & 3 != 0)
// build: hipcc --offload-arch=gfx950 -O2 tools/probe_buffer.hip -o tools/probe_buffer
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <vector>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ i32x4 make_srd(const void *p, uint32_t bytes)
{
    const uint64_t a = (uint64_t)p;
    i32x4 r;
    r.x = __builtin_amdgcn_readfirstlane((int)(uint32_t)a);
    r.y = __builtin_amdgcn_readfirstlane((int)(uint32_t)(a >> 32) & 0xffff);
    r.z = __builtin_amdgcn_readfirstlane((int)bytes);
    r.w = 0x00020000;
    return r;
}

__global__ void t1_store(uint8_t *buf, uint32_t records, uint32_t soff, uint32_t vbase)
{
    const i32x4 srd = make_srd(buf, records);
    const uint32_t lane = threadIdx.x;
    const u32x4 v = {0x11111111u * (lane & 15), lane, 0xdeadbeefu, soff};
    const uint32_t voff = vbase + 16 * lane;
    const uint32_t s = __builtin_amdgcn_readfirstlane(soff);
    asm volatile("s_nop 4\n\tbuffer_store_dwordx4 %0, %1, %2, %3 offen\n\ts_waitcnt vmcnt(0)" ::"v"(v), "v"(voff), "s"(srd), "s"(s) : "memory");
}

__global__ void t2_dma(const uint8_t *src, uint32_t records, uint32_t soff, uint32_t *out)
{
    __shared__ __attribute__((aligned(16))) uint32_t lds[1024];   // 4 KiB
    for (int i = threadIdx.x; i < 1024; i += 64) lds[i] = 0xAAAAAAAAu;
    __syncthreads();
    const i32x4 srd = make_srd(src, records);
    const uint32_t lane = threadIdx.x;
    const uint32_t voff = 16 * (lane ^ 1);   // swap neighbouring lanes' sources: the destination stays lane-linear
    const uint32_t s = __builtin_amdgcn_readfirstlane(soff);
    const uint32_t l = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint32_t *)lds);
    asm volatile("s_mov_b32 m0, %3\n\ts_nop 4\n\t"
                 "buffer_load_dwordx4 %0, %1, %2 offen lds\n\t"
                 "buffer_load_dwordx4 %0, %1, %2 offen offset:1024 lds\n\t"
                 "s_waitcnt vmcnt(0)" ::"v"(voff), "s"(srd), "s"(s), "s"(l) : "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 1024; i += 64) out[i] = lds[i];
}

__global__ void t3_unaligned(const uint8_t *src, uint8_t *dst, int shift, uint32_t *out)
{
    const int lane = threadIdx.x;
    const uint8_t *p = src + shift + 32 * lane;
    const uint32_t a = *reinterpret_cast<const uint32_t *>(p);
    const uint2 b = *reinterpret_cast<const uint2 *>(p + 4);
    const uint4 c = *reinterpret_cast<const uint4 *>(p + 12);
    out[8 * lane + 0] = a; out[8 * lane + 1] = b.x; out[8 * lane + 2] = b.y;
    out[8 * lane + 3] = c.x; out[8 * lane + 4] = c.y; out[8 * lane + 5] = c.z; out[8 * lane + 6] = c.w;
    *reinterpret_cast<uint4 *>(dst + shift + 32 * lane) = make_uint4(a, b.x, b.y, c.x);
    // the same through a buffer resource
    const i32x4 srd = make_srd(src, 4096);
    uint32_t d;
    const uint32_t voff = shift + 32 * lane + 28;
    asm volatile("s_nop 4\n\tbuffer_load_dword %0, %1, %2, 0 offen\n\ts_waitcnt vmcnt(0)" : "=v"(d) : "v"(voff), "s"(srd) : "memory");
    out[8 * lane + 7] = d;
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

int main()
{
    uint8_t *buf; uint32_t *out;
    CK(hipMalloc(&buf, 1 << 16)); CK(hipMalloc(&out, 1 << 16));
    std::vector<uint8_t> h(1 << 16);
    // ---- T1 ----
    printf("T1 buffer_store_dwordx4, num_records = 1024, 64 lanes x 16 B at voffset = vbase + 16 lane\n");
    const uint32_t cases[][2] = {{0, 0}, {512, 0}, {1008, 0}, {1024, 0}, {0, 512}, {256, 512}, {0, 0x80000000u}, {512, 1016}};
    for (auto &c : cases) {
        CK(hipMemset(buf, 0, 1 << 16));
        hipLaunchKernelGGL(t1_store, dim3(1), dim3(64), 0, 0, buf, 1024u, c[0], c[1]);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(h.data(), buf, 1 << 16, hipMemcpyDeviceToHost));
        int first = -1, last = -1, n = 0;
        for (int i = 0; i < (1 << 16); ++i) if (h[i]) { if (first < 0) first = i; last = i; ++n; }
        printf("   soffset %5u vbase %10u: nonzero bytes %5d, first %6d, last %6d  (in-range by voffset+soffset rule: bytes [%u, 1024))\n", c[0], c[1], n, first, last, c[0] + c[1]);
    }
    // ---- T2 ----
    printf("T2 buffer_load_dwordx4 ... lds, two instructions (offset:0 and offset:1024), voffset = 16 (lane ^ 1), M0 = lds base\n");
    for (int i = 0; i < 8192; ++i) h[i] = (uint8_t)(i / 16);   // chunk index (mod 256)
    CK(hipMemcpy(buf, h.data(), 8192, hipMemcpyHostToDevice));
    const uint32_t c2[][2] = {{8192, 0}, {8192, 2048}, {1536, 0}, {1536, 1024}};
    for (auto &c : c2) {
        hipLaunchKernelGGL(t2_dma, dim3(1), dim3(64), 0, 0, buf, c[0], c[1], out);
        CK(hipDeviceSynchronize());
        std::vector<uint32_t> r(1024);
        CK(hipMemcpy(r.data(), out, 4096, hipMemcpyDeviceToHost));
        printf("   num_records %5u soffset %5u: LDS chunk -> source chunk:", c[0], c[1]);
        for (int ch : {0, 1, 2, 63, 64, 65, 127, 128, 200}) {
            const uint32_t v = r[4 * ch];
            if (v == 0xAAAAAAAAu) printf("  %d:untouched", ch);
            else printf("  %d:%u%s", ch, v & 0xff, (v == 0 && ch != 0) ? "(zero)" : "");
        }
        int untouched = 0, zero = 0;
        for (int ch = 0; ch < 128; ++ch) { untouched += r[4 * ch] == 0xAAAAAAAAu; zero += r[4 * ch] == 0 && r[4 * ch + 1] == 0; }
        printf("   [of the 128 destination chunks: %d untouched, %d zero]\n", untouched, zero);
    }
    // ---- T3 ----
    printf("T3 unaligned accesses (global_load_dword/x2/x4, global_store_dwordx4, buffer_load_dword) at address & 3 = shift\n");
    for (int i = 0; i < 8192; ++i) h[i] = (uint8_t)(i * 7 + 3);
    CK(hipMemcpy(buf, h.data(), 8192, hipMemcpyHostToDevice));
    for (int shift = 0; shift < 4; ++shift) {
        CK(hipMemset(buf + 16384, 0, 4096));
        hipLaunchKernelGGL(t3_unaligned, dim3(1), dim3(64), 0, 0, buf, buf + 16384, shift, out);
        hipError_t e = hipDeviceSynchronize();
        if (e != hipSuccess) { printf("   shift %d: FAULT %s\n", shift, hipGetErrorString(e)); return 1; }
        std::vector<uint32_t> r(512); std::vector<uint8_t> d(4096);
        CK(hipMemcpy(r.data(), out, 2048, hipMemcpyDeviceToHost));
        CK(hipMemcpy(d.data(), buf + 16384, 4096, hipMemcpyDeviceToHost));
        int bad_load = 0, bad_store = 0;
        for (int lane = 0; lane < 64; ++lane) {
            const uint8_t *p = h.data() + shift + 32 * lane;
            uint32_t want[8]; memcpy(want, p, 32);
            for (int k = 0; k < 8; ++k) bad_load += r[8 * lane + k] != want[k];
            bad_store += memcmp(d.data() + shift + 32 * lane, p, 16) != 0;
        }
        printf("   shift %d: wrong loaded dwords %d / 512, wrong stored chunks %d / 64\n", shift, bad_load, bad_store);
    }
    return 0;
}
