// probe_xcd.hip -- the assumptions an XCD-local producer / consumer pipeline inside ONE kernel rests on:
//  (1) HW_REG_XCC_ID tells a wave which XCD it runs on; how a grid's workgroups spread over the XCDs;
//  (2) data a wave wrote with plain stores, followed by s_waitcnt vmcnt(0) and a relaxed atomic on a
//      flag, is visible to waves of the SAME XCD that poll the flag and read the data with sc1 loads
//      (L1-bypassing; the XCD's L2 is the point of coherence) -- including LDS-DMA loads;
//  (3) device-scope atomics on one address are coherent across XCDs (tickets sum up).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstdlib>
constexpr int NT = 4096;        // producer tasks per XCD and round
constexpr int MAXX = 16;
struct Ctl { unsigned ticket[MAXX]; unsigned flag[MAXX][NT]; unsigned err, timeouts, total, hist[MAXX]; };

__device__ inline unsigned xcc_id() { unsigned x; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x)); return x & 15u; }
__device__ inline unsigned pattern(unsigned round, unsigned xcc, unsigned p, unsigned i) { return (round * 2654435761u) ^ (xcc << 28) ^ (p << 12) ^ i; }

template <bool DMA>
__global__ __launch_bounds__(256) void k_stress(Ctl *c, unsigned *buf, unsigned round, int stage)
{
    __shared__ unsigned lds[4][2048];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned xcc = __builtin_amdgcn_readfirstlane(xcc_id());
    if (threadIdx.x == 0) atomicAdd(&c->hist[xcc], 1u);
    if (stage == 0) return;
    if (stage == 7) { if (lane == 0) __hip_atomic_fetch_add(&c->total, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); return; }
    if (stage == 8) { if (lane == 0) for (int i = 0; i < 16; ++i) __hip_atomic_fetch_add(&c->total, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); return; }
    if (stage == 9) { unsigned v = 0; if (lane == 0) for (int i = 0; i < 16; ++i) v += __hip_atomic_fetch_add(&c->total, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); if (v == 0xffffffffu) c->err = v; return; }
    for (;;) {
        unsigned t = 0;
        if (lane == 0) t = __hip_atomic_fetch_add(&c->ticket[xcc], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        t = __builtin_amdgcn_readfirstlane(t);
        if (t >= 2 * NT) break;
        const unsigned p = t >> 1;
        unsigned *mine = buf + ((size_t)xcc * NT + p) * 2048;
        if (stage == 1) continue;
        if (stage == 6) { if (lane == 0) __hip_atomic_fetch_add(&c->total, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); continue; }
        if ((t & 1) == 0 || stage == 2 || stage == 4 || stage == 5) {
            for (int i = 0; i < 32; ++i) mine[64 * i + lane] = pattern(round, xcc, p, 64 * i + lane);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (lane == 0 && stage != 4) __hip_atomic_fetch_add(&c->flag[xcc][p], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            int spins = 0;
            while (__hip_atomic_load(&c->flag[xcc][p], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < round + 1) {
                __builtin_amdgcn_s_sleep(2);
                if (++spins > (1 << 16)) { if (lane == 0) atomicAdd(&c->timeouts, 1u); break; }
            }
            unsigned bad = 0;
            if (DMA) {
                const unsigned ldsaddr = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned *)lds[wave]);
                for (int i = 0; i < 32; ++i)
                    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dword %0, off sc1" ::"v"(mine + 64 * i + lane), "s"(__builtin_amdgcn_readfirstlane(ldsaddr + 256 * i)) : "memory");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                for (int i = 0; i < 32; ++i) bad += lds[wave][64 * i + lane] != pattern(round, xcc, p, 64 * i + lane);
            } else {
                for (int i = 0; i < 32; ++i) {
                    unsigned v; asm volatile("global_load_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(mine + 64 * i + lane) : "memory");
                    bad += v != pattern(round, xcc, p, 64 * i + lane);
                }
            }
            if (bad) atomicAdd(&c->err, bad);
        }
        if (lane == 0 && stage < 4) __hip_atomic_fetch_add(&c->total, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}
int main(int argc, char **argv)
{
    const int stage = argc > 1 ? atoi(argv[1]) : 3;
    setvbuf(stdout, nullptr, _IONBF, 0);
    Ctl *c; unsigned *buf;
    (void)hipMalloc(&c, sizeof(Ctl)); (void)hipMemset(c, 0, sizeof(Ctl));
    (void)hipMalloc(&buf, (size_t)MAXX * NT * 2048 * 4);
    std::vector<unsigned char> hostmem(sizeof(Ctl)); Ctl *h = reinterpret_cast<Ctl *>(hostmem.data());
    for (int mode = 0; mode < 2; ++mode) {
        (void)hipMemset(c, 0, sizeof(Ctl));
        hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        (void)hipEventRecord(e0);
        const int rounds = 10;
        for (unsigned r = 0; r < rounds; ++r) {
            (void)hipMemsetAsync(c->ticket, 0, sizeof(c->ticket), 0);
            if (mode) hipLaunchKernelGGL(k_stress<true>, dim3(768), dim3(256), 0, 0, c, buf, r, stage);
            else hipLaunchKernelGGL(k_stress<false>, dim3(768), dim3(256), 0, 0, c, buf, r, stage);
            hipError_t e = hipDeviceSynchronize(); printf("mode %d round %u: %s\n", mode, r, hipGetErrorString(e));
        }
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        (void)hipMemcpy(h, c, sizeof(Ctl), hipMemcpyDeviceToHost);
        printf("%s reads: %d rounds, %.2f ms, tasks done %u (cross-XCD atomic; expected %u x XCDs), mismatching dwords %u, spin time-outs %u\n",
               mode ? "LDS-DMA sc1" : "global_load sc1", rounds, ms, h->total, 2 * NT * rounds, h->err, h->timeouts);
        printf("  workgroups per XCC_ID (768 per launch):");
        int nx = 0; for (int i = 0; i < MAXX; ++i) if (h->hist[i]) { printf(" [%d] %u", i, h->hist[i] / rounds); ++nx; }
        printf("  -> %d XCDs, tasks/expected = %.3f\n", nx, (double)h->total / (2.0 * NT * rounds * nx));
    }
    return 0;
}
