// probe_cvt_round.hip -- does v_cvt_pk_u8_f32 follow the wave's f32 rounding mode (MODE.FP_ROUND)?  If it does, "floor, then
// saturate + pack" (v_floor_f32 + v_cvt_pk_u8_f32, two half-rate instructions per sample) is ONE instruction in round-down mode.
#include <hip/hip_runtime.h>
#include <cstdio>

template <int RM>
__global__ void k(const float *in, unsigned *out, float *sum, int n)
{
    const int i = threadIdx.x;
    if (i >= n) return;
    float x = in[i];
    unsigned r = 0;
    float s;
    // the mode is switched around the conversion only; an f32 add under the same mode shows that the switch took effect
    asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 0, 2), %3\n\t"
                 "v_cvt_pk_u8_f32 %0, %2, 0, %0\n\t"
                 "v_add_f32 %1, %2, %4\n\t"
                 "s_setreg_imm32_b32 hwreg(HW_REG_MODE, 0, 2), 0"
                 : "+v"(r), "=v"(s) : "v"(x), "n"(RM), "v"(1.0e-9f));
    out[i] = r & 0xff;
    sum[i] = s;
}

int main()
{
    const float h[] = {-1.5f, -0.5f, -0.25f, 0.0f, 0.25f, 0.5f, 0.75f, 1.0f, 1.5f, 2.5f, 3.5f, 126.5f, 127.5f, 127.99999f, 128.0f,
                       254.5f, 254.99998f, 255.0f, 255.25f, 255.5f, 255.99998f, 256.0f, 1000.0f, -1000.0f};
    const int n = sizeof(h) / sizeof(h[0]);
    float *d; unsigned *o; float *s;
    (void)hipMalloc(&d, sizeof(h)); (void)hipMalloc(&o, n * 4); (void)hipMalloc(&s, n * 4);
    (void)hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
    unsigned r[4][64]; float t[4][64];
    hipLaunchKernelGGL(k<0>, dim3(1), dim3(64), 0, 0, d, o, s, n); (void)hipMemcpy(r[0], o, n * 4, hipMemcpyDeviceToHost); (void)hipMemcpy(t[0], s, n * 4, hipMemcpyDeviceToHost);
    hipLaunchKernelGGL(k<1>, dim3(1), dim3(64), 0, 0, d, o, s, n); (void)hipMemcpy(r[1], o, n * 4, hipMemcpyDeviceToHost); (void)hipMemcpy(t[1], s, n * 4, hipMemcpyDeviceToHost);
    hipLaunchKernelGGL(k<2>, dim3(1), dim3(64), 0, 0, d, o, s, n); (void)hipMemcpy(r[2], o, n * 4, hipMemcpyDeviceToHost); (void)hipMemcpy(t[2], s, n * 4, hipMemcpyDeviceToHost);
    hipLaunchKernelGGL(k<3>, dim3(1), dim3(64), 0, 0, d, o, s, n); (void)hipMemcpy(r[3], o, n * 4, hipMemcpyDeviceToHost); (void)hipMemcpy(t[3], s, n * 4, hipMemcpyDeviceToHost);
    printf("%14s  %8s %8s %8s %8s   (x + 1e-9 under the mode: nearest / +inf / -inf / zero)\n", "x", "nearest", "+inf", "-inf", "zero");
    for (int i = 0; i < n; ++i)
        printf("%14.6f  %8u %8u %8u %8u   %.9g %.9g %.9g %.9g\n", h[i], r[0][i], r[1][i], r[2][i], r[3][i], t[0][i], t[1][i], t[2][i], t[3][i]);
    return 0;
}
