// probe_place.hip -- round 6: where do the workgroups and waves of the encode kernel's launch (1024 x 256 work-items, 25 KB of LDS: four
// per CU) land?  Every wave records HW_ID / XCC_ID; reported: which workgroups share a CU, which SIMD each wave of a workgroup gets.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
#include <vector>
#include <algorithm>
__global__ __launch_bounds__(256) void k(unsigned *rec, float *sink)
{
    __shared__ float buf[6400];
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    buf[threadIdx.x] = (float)hw;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) { rec[2 * (blockIdx.x * 4 + (threadIdx.x >> 6))] = hw; rec[2 * (blockIdx.x * 4 + (threadIdx.x >> 6)) + 1] = xcc & 15u; }
    float x = buf[(threadIdx.x * 7) % 6400];
    for (int i = 0; i < 20000; ++i) x = x * 1.0000001f + 0.5f;
    if (x == 12345.0f) sink[0] = x;
}
int main()
{
    const int grid = 1024;
    unsigned *d; float *sink;
    (void)hipMalloc(&d, grid * 4 * 8); (void)hipMalloc(&sink, 4);
    std::vector<unsigned> h(grid * 8);
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(k, dim3(grid), dim3(256), 0, 0, d, sink);
        (void)hipDeviceSynchronize();
        (void)hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
        std::map<unsigned, std::vector<int>> cu;   // key: xcc, se, sh, cu
        int distinct4 = 0; std::map<std::string, int> patterns;
        for (int b = 0; b < grid; ++b) {
            unsigned simds = 0; char pat[8];
            for (int w = 0; w < 4; ++w) { const unsigned hw = h[2 * (b * 4 + w)]; const unsigned s = (hw >> 4) & 3; simds |= 1u << s; pat[w] = '0' + s; }
            pat[4] = 0; patterns[pat]++;
            distinct4 += simds == 15u;
            const unsigned hw = h[2 * (b * 4)], xcc = h[2 * (b * 4) + 1];
            const unsigned key = (xcc << 16) | (hw & 0xff00u);   // CU_ID [11:8], SH_ID [12], SE_ID [15:13]
            cu[key].push_back(b);
        }
        printf("launch %d: %zu distinct (xcc, se, sh, cu); workgroups with four distinct SIMDs: %d of %d\n", rep, cu.size(), distinct4, grid);
        printf("  wave -> SIMD patterns:"); for (auto &p : patterns) printf(" %s x%d", p.first.c_str(), p.second); printf("\n");
        int same256 = 0, n = 0; std::map<int, int> per_cu;
        for (auto &c : cu) {
            per_cu[(int)c.second.size()]++;
            bool all = true; for (int b : c.second) all = all && (b % 256 == c.second[0] % 256);
            same256 += all; ++n;
        }
        printf("  workgroups per CU:"); for (auto &p : per_cu) printf(" %d x%d", p.first, p.second);
        printf("\n  CUs whose workgroups are all congruent mod 256: %d of %d\n  examples:", same256, n);
        int shown = 0;
        for (auto &c : cu) { if (shown++ >= 6) break; printf(" [%05x:", c.first); for (int b : c.second) { printf(" %d(", b); for (int w = 0; w < 4; ++w) printf("%u", (h[2 * (b * 4 + w)] >> 4) & 3); printf(")"); } printf("]"); }
        printf("\n");
    }
    return 0;
}
