// probe_store_pattern.hip -- does the SHAPE of k_luma_fused's output strips matter to HBM?
// Writes an 8192 x 8192 RGB8 image (201 MB) with the kernel's persistent-wave schedule and nt
// 16-byte stores, once as 32 x 2-block strips (two 768-byte runs, 8 rows apart, per stored row
// pair) and once as 64 x 1-block strips (one 1536-byte run per stored row).  No arithmetic.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned int u4 __attribute__((ext_vector_type(4)));

template <int SHAPE>   // 0: 256 x 16 px strips, 1: 512 x 8 px strips
__global__ __launch_bounds__(256) void k_store(unsigned char *out, int W, int H)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nwaves = gridDim.x * 4;
    const size_t pitch = (size_t)W * 3;
    const u4 v = {(unsigned)lane, blockIdx.x, 3u, 4u};
    if (SHAPE == 0) {
        const int tx = W / 256, total = tx * (H / 16);
        for (int s = blockIdx.x * 4 + wave; s < total; s += nwaves) {
            const int sy = s / tx, sx = s - sy * tx;
            unsigned char *base = out + (size_t)(16 * sy) * pitch + (size_t)sx * 768;
            const int sg0 = lane >= 48 ? 1 : 0, j0 = lane - 48 * sg0;
            for (int y = 0; y < 8; ++y) {
                __builtin_nontemporal_store(v, (u4 *)(base + (size_t)(y + 8 * sg0) * pitch + 16 * j0));
                if (lane < 32) __builtin_nontemporal_store(v, (u4 *)(base + (size_t)(y + 8) * pitch + 16 * (16 + lane)));
            }
        }
    } else {
        const int tx = W / 512, total = tx * (H / 8);
        for (int s = blockIdx.x * 4 + wave; s < total; s += nwaves) {
            const int sy = s / tx, sx = s - sy * tx;
            unsigned char *base = out + (size_t)(8 * sy) * pitch + (size_t)sx * 1536;
            for (int y = 0; y < 8; ++y) {
                __builtin_nontemporal_store(v, (u4 *)(base + (size_t)y * pitch + 16 * lane));
                if (lane < 32) __builtin_nontemporal_store(v, (u4 *)(base + (size_t)y * pitch + 16 * (64 + lane)));
            }
        }
    }
}

int main()
{
    const int W = 8192, H = 8192;
    const size_t bytes = (size_t)W * H * 3;
    unsigned char *buf[4];
    for (auto &b : buf) (void)hipMalloc(&b, bytes);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep)
        for (int shape = 0; shape < 2; ++shape) {
            (void)hipEventRecord(e0);
            for (int i = 0; i < 20; ++i) {
                if (shape == 0) hipLaunchKernelGGL(k_store<0>, dim3(768), dim3(256), 0, 0, buf[i & 3], W, H);
                else hipLaunchKernelGGL(k_store<1>, dim3(768), dim3(256), 0, 0, buf[i & 3], W, H);
            }
            (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1);
            printf("%s strips: %6.1f us per image, %5.0f GB/s\n", shape ? "64 x 1" : "32 x 2", ms / 20 * 1e3, bytes / (ms / 20) / 1e6);
        }
    return 0;
}
