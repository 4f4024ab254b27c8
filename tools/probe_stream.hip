// probe_stream.hip -- the memory side of the fused 4:2:0 decode alone: what does HBM sustain for
// the step's REAL footprint (read 201 MB of coefficient blocks, write 201 MB of RGB rows) as a
// function of the tile shape, the number of resident workgroups, the load flavour (plain 16-byte
// loads or LDS-DMA, `nt` or not) and the store flavour?  No arithmetic beyond an XOR that keeps the
// loads alive.  VERDICT r01 task 1(b): sweep the grid and use whole-line stores / LDS-DMA reads.
//
// A tile is BW x BH luma blocks of an 8192 x 8192 image; persistent waves take tiles round-robin in
// row-major order (consecutive waves = horizontally adjacent tiles), exactly like the kernels.
//   reads : BH runs of BW x 128 B from the luma plane (pitch 1024 x 128 B) + BW x BH x 64 B of chroma
//   writes: 8 BH pixel rows of BW x 24 B each (pitch 24 576 B)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef unsigned int u4 __attribute__((ext_vector_type(4)));

constexpr int W = 8192, H = 8192, UX = W / 8, UY = H / 8;

__device__ __forceinline__ void lds_dma16(const void *g, unsigned lds, bool nt)
{
    if (nt) asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off nt" ::"v"(g), "s"(lds) : "memory");
    else asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(g), "s"(lds) : "memory");
}

// LOAD: 0 plain global_load_dwordx4, 1 LDS-DMA, 2 LDS-DMA nt;  STORE: 0 plain, 1 nt
template <int BW, int BH, int LOAD, int STORE>
__global__ __launch_bounds__(256) void k_stream(const unsigned char *luma, const unsigned char *chroma, unsigned char *out)
{
    __shared__ __attribute__((aligned(16))) u4 buf[4][LOAD ? 768 : 1];   // 12 KiB per wave for the DMA variants
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nwaves = gridDim.x * 4;
    constexpr int TX = UX / BW, TY = UY / BH;
    constexpr int RUN = BW * 128;                  // bytes per luma run
    constexpr int CH = BW * BH * 64;               // chroma bytes per tile
    constexpr int ROWB = BW * 24;                  // bytes per pixel row of a tile
    const size_t pitch = (size_t)W * 3;
    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) u4 *)buf[wave]);
    for (int t = blockIdx.x * 4 + wave; t < TX * TY; t += nwaves) {
        const int ty = t / TX, tx = t - ty * TX;
        u4 acc = {0, 0, 0, 0};
        if (LOAD == 0) {
#pragma unroll
            for (int r = 0; r < BH; ++r) {
                const unsigned char *src = luma + ((size_t)(ty * BH + r) * UX + (size_t)tx * BW) * 128;
#pragma unroll
                for (int o = 0; o < RUN; o += 1024) acc ^= *(const u4 *)(src + o + 16 * lane);
            }
            const unsigned char *csrc = chroma + (size_t)t * CH;
#pragma unroll
            for (int o = 0; o < CH; o += 1024) acc ^= *(const u4 *)(csrc + o + 16 * lane);
        } else {
            int slot = 0;
#pragma unroll
            for (int r = 0; r < BH; ++r) {
                const unsigned char *src = luma + ((size_t)(ty * BH + r) * UX + (size_t)tx * BW) * 128;
#pragma unroll
                for (int o = 0; o < RUN; o += 1024) lds_dma16(src + o + 16 * lane, lds0 + 1024 * slot++, LOAD == 2);
            }
            const unsigned char *csrc = chroma + (size_t)t * CH;
#pragma unroll
            for (int o = 0; o < CH; o += 1024) lds_dma16(csrc + o + 16 * lane, lds0 + 1024 * slot++, LOAD == 2);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
            for (int i = 0; i < (BH * RUN + CH) / 1024; ++i) acc ^= buf[wave][64 * i + lane];
        }
        // Every lane's loads must really be issued: the stores below only use lanes 0 .. CPR-1, and without this pin the
        // compiler sinks the plain loads into that branch -- 24 or 48 of the 64 lanes load, and the table reads 8.0 / 6.4
        // TB/s for bytes that were never moved (round 2's first version of this probe did exactly that).
        asm volatile("" : "+v"(acc.x), "+v"(acc.y), "+v"(acc.z), "+v"(acc.w));
        unsigned char *base = out + (size_t)(8 * BH * ty) * pitch + (size_t)tx * ROWB;
        constexpr int CPR = ROWB / 16;             // 16-byte chunks per row
        if constexpr (STORE == 2) {
            // k_luma_fused's grouping: per pixel row y of every block row, the BH segments are BH * CPR = 96 chunks;
            // a lane stores chunk `lane` and lanes 0..31 also chunk 64 + lane (two nt instructions per y)
#pragma unroll
            for (int y = 0; y < 8; ++y) {
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    const int u = 64 * k + lane, sg = u / CPR, j = u - CPR * sg;
                    if (u < BH * CPR) __builtin_nontemporal_store(acc, (u4 *)(base + (size_t)(8 * sg + y) * pitch + 16 * j));
                }
            }
            continue;
        }
        if constexpr (STORE == 3) {
            // one instruction per row segment like STORE 1, but in the kernel's ORDER: pixel row y of every block row
            // (rows y, y + 8, y + 16, ...), then y + 1 ...
#pragma unroll
            for (int y = 0; y < 8; ++y)
#pragma unroll
                for (int sg = 0; sg < BH; ++sg)
                    if (lane < CPR) __builtin_nontemporal_store(acc, (u4 *)(base + (size_t)(8 * sg + y) * pitch + 16 * lane));
            continue;
        }
#pragma unroll
        for (int y = 0; y < 8 * BH; ++y) {
#pragma unroll
            for (int c = 0; c < CPR; c += 64) {
                if (c + lane < CPR) {
                    u4 *p = (u4 *)(base + (size_t)y * pitch + 16 * (c + lane));
                    if (STORE) __builtin_nontemporal_store(acc, p); else *p = acc;
                }
            }
        }
    }
}


// The fused kernel's SCHEDULE with plain coalesced loads: the next tile's 12 loads are issued BEFORE this tile's stores
// (register prefetch, as k_luma_fused does with its coefficient buffer), the stores follow one pixel row at a time with an
// LDS round trip in between (MODE bit 0), optionally with a pause of `gap` s_sleep units between rows (MODE bit 1).
template <int BW, int BH, int MODE>
__global__ __launch_bounds__(256) void k_sched(const unsigned char *luma, const unsigned char *chroma, unsigned char *out)
{
    __shared__ __attribute__((aligned(16))) u4 stage[4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nwaves = gridDim.x * 4;
    constexpr int TX = UX / BW, TY = UY / BH, RUN = BW * 128, CH = BW * BH * 64, ROWB = BW * 24, CPR = ROWB / 16;
    constexpr int NL = (BH * RUN + CH) / 1024;
    const size_t pitch = (size_t)W * 3;
    u4 nxt[NL];
    auto fetch = [&](int t) {
        const int ty = t / TX, tx = t - ty * TX;
        int k = 0;
#pragma unroll
        for (int r = 0; r < BH; ++r) {
            const unsigned char *src = luma + ((size_t)(ty * BH + r) * UX + (size_t)tx * BW) * 128;
#pragma unroll
            for (int o = 0; o < RUN; o += 1024) nxt[k++] = *(const u4 *)(src + o + 16 * lane);
        }
        const unsigned char *csrc = chroma + (size_t)t * CH;
#pragma unroll
        for (int o = 0; o < CH; o += 1024) nxt[k++] = *(const u4 *)(csrc + o + 16 * lane);
    };
    int t = blockIdx.x * 4 + wave;
    if (t >= TX * TY) return;
    fetch(t);
    for (; t < TX * TY; t += nwaves) {
        const int ty = t / TX, tx = t - ty * TX;
        u4 acc = {0, 0, 0, 0};
#pragma unroll
        for (int k = 0; k < NL; ++k) acc ^= nxt[k];
        asm volatile("" : "+v"(acc.x), "+v"(acc.y), "+v"(acc.z), "+v"(acc.w));   // all 64 lanes load (see k_stream)
        if (!(MODE & 8)) { if (t + nwaves < TX * TY) fetch(t + nwaves); }
        unsigned char *base = out + (size_t)(8 * BH * ty) * pitch + (size_t)tx * ROWB;
#pragma unroll
        for (int y = 0; y < 8; ++y) {
            if (MODE & 1) {   // an LDS round trip per pixel row, like the kernel's store staging (wave-private, no race: own slot)
                volatile __attribute__((address_space(3))) u4 *slot = (volatile __attribute__((address_space(3))) u4 *)&stage[wave][lane];
                *slot = acc;
                __builtin_amdgcn_wave_barrier();
                const u4 back = *slot;
                acc ^= back; acc ^= back;   // keeps the read alive and acc unchanged
                acc.x += back.y & 1u;
            }
#pragma unroll
            for (int sg = 0; sg < BH; ++sg)
                if (lane < CPR) __builtin_nontemporal_store(acc, (u4 *)(base + (size_t)(8 * sg + y) * pitch + 16 * lane));
            if (MODE & 2) __builtin_amdgcn_s_sleep(8);
        }
        if (MODE & 4) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // drain: this tile's stores (and the prefetch) before going on
        if (MODE & 8) { if (t + nwaves < TX * TY) fetch(t + nwaves); }    // the prefetch issued AFTER the stores instead of before
    }
}

// Late prefetch (the next tile's loads after this tile's stores) with the tile's 8 BH store instructions issued in GROUPS of G,
// an LDS round trip (ds_write + ds_read + wait, like the kernel's store staging) before every group: how long must a burst be?
template <int BW, int BH, int G>
__global__ __launch_bounds__(256) void k_group(const unsigned char *luma, const unsigned char *chroma, unsigned char *out)
{
    __shared__ __attribute__((aligned(16))) u4 stage[4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nwaves = gridDim.x * 4;
    constexpr int TX = UX / BW, TY = UY / BH, RUN = BW * 128, CH = BW * BH * 64, ROWB = BW * 24, CPR = ROWB / 16;
    constexpr int NL = (BH * RUN + CH) / 1024;
    const size_t pitch = (size_t)W * 3;
    u4 nxt[NL];
    auto fetch = [&](int t) {
        const int ty = t / TX, tx = t - ty * TX;
        int k = 0;
#pragma unroll
        for (int r = 0; r < BH; ++r) {
            const unsigned char *src = luma + ((size_t)(ty * BH + r) * UX + (size_t)tx * BW) * 128;
#pragma unroll
            for (int o = 0; o < RUN; o += 1024) nxt[k++] = *(const u4 *)(src + o + 16 * lane);
        }
        const unsigned char *csrc = chroma + (size_t)t * CH;
#pragma unroll
        for (int o = 0; o < CH; o += 1024) nxt[k++] = *(const u4 *)(csrc + o + 16 * lane);
    };
    int t = blockIdx.x * 4 + wave;
    if (t >= TX * TY) return;
    fetch(t);
    volatile __attribute__((address_space(3))) u4 *slot = (volatile __attribute__((address_space(3))) u4 *)&stage[wave][lane];
    for (; t < TX * TY; t += nwaves) {
        const int ty = t / TX, tx = t - ty * TX;
        u4 acc = {0, 0, 0, 0};
#pragma unroll
        for (int k = 0; k < NL; ++k) acc ^= nxt[k];
        asm volatile("" : "+v"(acc.x), "+v"(acc.y), "+v"(acc.z), "+v"(acc.w));   // all 64 lanes load (see k_stream)
        unsigned char *base = out + (size_t)(8 * BH * ty) * pitch + (size_t)tx * ROWB;
#pragma unroll
        for (int i = 0; i < 8 * BH; ++i) {   // store i: pixel row i / BH of segment i % BH (the kernel's order)
            if (i % (G & 63) == 0) {
                if (G & 256) {   // a pause instead of the LDS round trip
                    __builtin_amdgcn_s_sleep(4);
                } else if (G & 128) {   // the LDS round trip, but the stores do not depend on it
                    *slot = acc;
                    __builtin_amdgcn_wave_barrier();
                    const u4 back = *slot;
                    if (back.y == 0x12345u && back.x == 77u) nxt[0].x = 1;
                } else if (G & 64) {   // dependent VALU chain instead (about the latency of an LDS round trip)
#pragma unroll
                    for (int k = 0; k < 32; ++k) acc.x = acc.x * 3u + 1u;
                } else {
                    *slot = acc;
                    __builtin_amdgcn_wave_barrier();
                    const u4 back = *slot;
                    acc.x += back.y & 1u;
                }
            }
            if (lane < CPR) __builtin_nontemporal_store(acc, (u4 *)(base + (size_t)(8 * (i % BH) + i / BH) * pitch + 16 * lane));
        }
        if (t + nwaves < TX * TY) fetch(t + nwaves);
    }
}

// pseudo-random fill (PROBE_RANDOM=1): is the rate data-dependent?
__global__ void k_fill(unsigned *p, size_t n, unsigned seed)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned x = (unsigned)i * 2654435761u + seed; x ^= x >> 15; x *= 2246822519u; x ^= x >> 13; x *= 3266489917u; x ^= x >> 16;
        p[i] = x;
    }
}

typedef void (*kfn)(const unsigned char *, const unsigned char *, unsigned char *);
struct Variant { const char *name; kfn fn; };

int main(int argc, char **argv)
{
    const size_t out_bytes = (size_t)W * H * 3, luma_bytes = (size_t)UX * UY * 128, chroma_bytes = luma_bytes / 2;
    const int ring = 8;
    unsigned char *luma[ring], *chroma[ring], *out[ring];
    for (int i = 0; i < ring; ++i) {
        (void)hipMalloc(&luma[i], luma_bytes); (void)hipMalloc(&chroma[i], chroma_bytes); (void)hipMalloc(&out[i], out_bytes);
        (void)hipMemset(luma[i], i + 1, luma_bytes); (void)hipMemset(chroma[i], i + 5, chroma_bytes);
        if (getenv("PROBE_RANDOM")) {
            hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, (unsigned *)luma[i], luma_bytes / 4, 17u * i + 1);
            hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, (unsigned *)chroma[i], chroma_bytes / 4, 29u * i + 3);
        }
    }
    (void)hipDeviceSynchronize();
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
#define V(bw, bh, ld, st) {#bw "x" #bh " load" #ld " store" #st, k_stream<bw, bh, ld, st>}
    std::vector<Variant> vs = {
        V(32, 2, 0, 1), V(32, 2, 2, 1), V(32, 2, 1, 1), V(32, 2, 2, 0), V(32, 2, 0, 0),
        V(64, 1, 0, 1), V(64, 1, 2, 1), V(64, 2, 0, 1), V(64, 2, 2, 1),
        V(16, 4, 0, 1), V(16, 4, 2, 1), V(128, 1, 0, 1), V(32, 4, 0, 1), V(32, 1, 0, 1), V(32, 1, 2, 1),
        V(16, 4, 0, 2), V(32, 2, 0, 2), V(16, 4, 2, 2), V(32, 2, 2, 2), V(16, 4, 0, 0),
        V(16, 4, 0, 3), V(32, 2, 0, 3), V(16, 4, 2, 3), V(8, 8, 0, 1), V(16, 8, 0, 1), V(16, 2, 0, 1),
        {"sched 16x4 prefetch", k_sched<16, 4, 0>}, {"sched 16x4 prefetch+lds", k_sched<16, 4, 1>}, {"sched 16x4 prefetch+lds+sleep", k_sched<16, 4, 3>},
        {"sched 32x2 prefetch", k_sched<32, 2, 0>}, {"sched 32x2 prefetch+lds", k_sched<32, 2, 1>},
        {"sched 16x4 prefetch+drain", k_sched<16, 4, 4>}, {"sched 16x4 late prefetch", k_sched<16, 4, 8>}, {"sched 16x4 late prefetch+drain", k_sched<16, 4, 12>},
        {"sched 32x2 prefetch+drain", k_sched<32, 2, 4>}, {"sched 32x2 late prefetch", k_sched<32, 2, 8>},
        {"sched 16x4 late prefetch+lds", k_sched<16, 4, 9>}, {"sched 32x2 late prefetch+lds", k_sched<32, 2, 9>},
        {"sched 16x4 late prefetch+lds+sleep", k_sched<16, 4, 11>},
        {"group 16x4 G=2", k_group<16, 4, 2>}, {"group 16x4 G=4", k_group<16, 4, 4>}, {"group 16x4 G=8", k_group<16, 4, 8>}, {"group 16x4 G=16", k_group<16, 4, 16>},
        {"group 16x4 G=32", k_group<16, 4, 32>}, {"group 32x2 G=2", k_group<32, 2, 2>}, {"group 32x2 G=4", k_group<32, 2, 4>}, {"group 32x2 G=8", k_group<32, 2, 8>},
        {"group 32x2 G=16", k_group<32, 2, 16>},
        {"gvar 16x4 sleep once", k_group<16, 4, 256 + 32>}, {"gvar 16x4 lds independent once", k_group<16, 4, 128 + 32>}, {"gvar 16x4 valu chain once", k_group<16, 4, 64 + 32>},
        {"gvar 16x4 sleep per 4", k_group<16, 4, 256 + 4>}, {"gvar 16x4 valu chain per 4", k_group<16, 4, 64 + 4>}};
    const int grids[] = {256, 512, 768, 1024, 1280, 1536, 2048, 3072};
    const double bytes = (double)(out_bytes + luma_bytes + chroma_bytes);
    printf("%-24s", "tile / flavour  \\  WGs");
    for (int g : grids) printf("%7d", g);
    printf("   (GB/s, 403 MB per pass; best of 3 x 16 passes over a ring of 8 buffer sets)\n");
    for (auto &v : vs) {
        if (argc > 1 && !strstr(v.name, argv[1])) continue;
        printf("%-32s", v.name);
        for (int g : grids) {
            float best = 1e9f;
            for (int rep = 0; rep < 3; ++rep) {
                (void)hipEventRecord(e0);
                for (int i = 0; i < 16; ++i) hipLaunchKernelGGL(v.fn, dim3(g), dim3(256), 0, 0, luma[i % ring], chroma[i % ring], out[i % ring]);
                (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
                float ms; (void)hipEventElapsedTime(&ms, e0, e1);
                if (ms / 16 < best) best = ms / 16;
            }
            printf("%7.0f", bytes / best / 1e6);
        }
        printf("\n"); fflush(stdout);
    }
    return 0;
}
