// probe_cvt_round2.hip -- hazards of switching MODE.FP_ROUND around v_cvt_pk_u8_f32: do f32 operations issued immediately
// BEFORE "s_setreg round-toward-zero" and immediately AFTER "s_setreg round-to-nearest" still round to nearest?
// a = 1 + 0.75 * 2^-23: RNE gives 1 + 2^-23 (0x3f800001), toward zero gives 1.0 (0x3f800000).
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ void k(unsigned *out, float one, float tiny, float x)
{
    float b0, b1, b2, b3, b4, b5, b6, b7, a0, a1, a2, a3;
    unsigned r = 0;
    asm volatile(
        "v_add_f32 %0, %13, %14\n\tv_add_f32 %1, %13, %14\n\tv_add_f32 %2, %13, %14\n\tv_add_f32 %3, %13, %14\n\t"
        "v_add_f32 %4, %13, %14\n\tv_add_f32 %5, %13, %14\n\tv_add_f32 %6, %13, %14\n\tv_add_f32 %7, %13, %14\n\t"
        "s_setreg_imm32_b32 hwreg(HW_REG_MODE, 0, 2), 3\n\t"
        "v_cvt_pk_u8_f32 %12, %15, 0, %12\n\t"
        "s_setreg_imm32_b32 hwreg(HW_REG_MODE, 0, 2), 0\n\t"
        "v_add_f32 %8, %13, %14\n\tv_add_f32 %9, %13, %14\n\tv_add_f32 %10, %13, %14\n\tv_add_f32 %11, %13, %14"
        : "=&v"(b0), "=&v"(b1), "=&v"(b2), "=&v"(b3), "=&v"(b4), "=&v"(b5), "=&v"(b6), "=&v"(b7),
          "=&v"(a0), "=&v"(a1), "=&v"(a2), "=&v"(a3), "+v"(r)
        : "v"(one), "v"(tiny), "v"(x));
    if (threadIdx.x == 0) {
        const float v[12] = {b0, b1, b2, b3, b4, b5, b6, b7, a0, a1, a2, a3};
        for (int i = 0; i < 12; ++i) out[i] = __float_as_uint(v[i]);
        out[12] = r;
    }
}

int main()
{
    unsigned *o; (void)hipMalloc(&o, 64);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, o, 1.0f, 0.75f * 1.1920929e-07f, 3.75f);
    unsigned h[13]; (void)hipMemcpy(h, o, sizeof(h), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 12; ++i) { printf("%s add %d: 0x%08x %s\n", i < 8 ? "before" : "after ", i, h[i], h[i] == 0x3f800001u ? "nearest" : "NOT nearest"); bad += h[i] != 0x3f800001u; }
    printf("cvt_pk_u8(3.75) under round-toward-zero: %u (3 expected)\n%s\n", h[12] & 0xff, bad == 0 && (h[12] & 0xff) == 3 ? "OK: the switch affects only the instructions between the two s_setreg" : "HAZARD");
    return 0;
}
