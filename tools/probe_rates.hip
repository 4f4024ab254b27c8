// probe_rates.hip -- issue rate of the VALU instructions the decode kernels lean on.
// Each kernel runs 8 independent chains of ONE instruction; rate in T lane-ops/s.
#include <hip/hip_runtime.h>
#include <cstdio>

#define CHAIN8(INSTR)                                                                         \
    asm volatile(INSTR(%0) "\n" INSTR(%1) "\n" INSTR(%2) "\n" INSTR(%3) "\n" INSTR(%4) "\n"     \
                 INSTR(%5) "\n" INSTR(%6) "\n" INSTR(%7)                                        \
                 : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) \
                 : "v"(a), "v"(b))

#define I_ADD(r) "v_add_f32 " #r ", " #r ", %8"
#define I_MUL(r) "v_mul_f32 " #r ", " #r ", %9"
#define I_FMA(r) "v_fma_f32 " #r ", " #r ", %9, %8"
#define I_FMAMK(r) "v_fmamk_f32 " #r ", " #r ", 0x3fb374bc, %8"
#define I_FLOOR(r) "v_floor_f32 " #r ", " #r
#define I_MED3(r) "v_med3_f32 " #r ", " #r ", %8, %9"
#define I_CVTI(r) "v_cvt_f32_i32 " #r ", " #r
#define I_CVTSDWA(r) "v_cvt_f32_i32_sdwa " #r ", sext(" #r ") dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1"
#define I_CVTUB(r) "v_cvt_f32_ubyte1 " #r ", " #r
#define I_CVTPK(r) "v_cvt_pk_u8_f32 " #r ", " #r ", 1, %8"
#define I_CVTU(r) "v_cvt_u32_f32 " #r ", " #r
#define I_LSHLOR(r) "v_lshl_or_b32 " #r ", " #r ", 8, %8"
#define I_PERM(r) "v_perm_b32 " #r ", " #r ", %8, %9"
#define I_MADU24(r) "v_mad_u32_u24 " #r ", " #r ", 3, %8"
#define I_PKMADU16(r) "v_pk_mad_u16 " #r ", " #r ", %9, %8"
#define I_PKFMA(r) "v_pk_fma_f32 " #r ", " #r ", %9, %8"

#define KERNEL(NAME, INSTR)                                                         \
    __global__ __launch_bounds__(256) void NAME(float *out, float a, float b, int iters) \
    {                                                                               \
        float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7; \
        for (int i = 0; i < iters; ++i) { CHAIN8(INSTR); CHAIN8(INSTR); }            \
        out[blockIdx.x * 256 + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;  \
    }

KERNEL(k_add, I_ADD) KERNEL(k_mul, I_MUL) KERNEL(k_fma, I_FMA) KERNEL(k_fmamk, I_FMAMK)
KERNEL(k_floor, I_FLOOR) KERNEL(k_med3, I_MED3) KERNEL(k_cvti, I_CVTI) KERNEL(k_cvtsdwa, I_CVTSDWA)
KERNEL(k_cvtub, I_CVTUB) KERNEL(k_cvtpk, I_CVTPK) KERNEL(k_cvtu, I_CVTU) KERNEL(k_lshlor, I_LSHLOR)
KERNEL(k_perm, I_PERM) KERNEL(k_madu24, I_MADU24) KERNEL(k_pkmadu16, I_PKMADU16)

typedef void (*kfn)(float *, float, float, int);

int main()
{
    float *d; (void)hipMalloc(&d, 256 * 2048 * 4);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    struct { const char *name; kfn fn; } ks[] = {
        {"v_add_f32", k_add}, {"v_mul_f32", k_mul}, {"v_fma_f32", k_fma}, {"v_fmamk_f32 (literal)", k_fmamk},
        {"v_floor_f32", k_floor}, {"v_med3_f32", k_med3}, {"v_cvt_f32_i32", k_cvti}, {"v_cvt_f32_i32_sdwa", k_cvtsdwa},
        {"v_cvt_f32_ubyte1", k_cvtub}, {"v_cvt_pk_u8_f32", k_cvtpk}, {"v_cvt_u32_f32", k_cvtu}, {"v_lshl_or_b32", k_lshlor},
        {"v_perm_b32", k_perm}, {"v_mad_u32_u24", k_madu24}, {"v_pk_mad_u16", k_pkmadu16}};
    const int iters = 10000, blocks = 2048;
    for (auto &k : ks) {
        float best = 1e9;
        for (int rep = 0; rep < 3; ++rep) {
            (void)hipEventRecord(e0);
            hipLaunchKernelGGL(k.fn, dim3(blocks), dim3(256), 0, 0, d, 1.0f, 1.0000001f, iters);
            (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1);
            if (ms < best) best = ms;
        }
        double ops = (double)blocks * 256 * iters * 16;
        printf("%-24s %8.3f ms  %6.2f T instr-lanes/s\n", k.name, best, ops / best / 1e9);
    }
    return 0;
}
