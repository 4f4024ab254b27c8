// probe_rates2.hip -- issue rate of further VALU candidates for the decode kernels (round 3): which instructions besides
// f32 add / mul / fma run at the full rate?  8 independent chains of ONE instruction, 8 waves per SIMD; T lane-ops/s.
#include <hip/hip_runtime.h>
#include <cstdio>

#define CHAIN8(INSTR)                                                                         \
    asm volatile(INSTR(%0) "\n" INSTR(%1) "\n" INSTR(%2) "\n" INSTR(%3) "\n" INSTR(%4) "\n"     \
                 INSTR(%5) "\n" INSTR(%6) "\n" INSTR(%7)                                        \
                 : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) \
                 : "v"(a), "v"(b))

#define I_ADD(r) "v_add_f32 " #r ", " #r ", %8"
#define I_DOT4(r) "v_dot4_u32_u8 " #r ", " #r ", %9, %8"
#define I_DOT2F(r) "v_dot2_f32_f16 " #r ", " #r ", %9, %8"
#define I_MAX(r) "v_max_f32 " #r ", " #r ", %8"
#define I_MOV(r) "v_mov_b32 " #r ", %8"
#define I_CNDMASK(r) "v_cndmask_b32 " #r ", " #r ", %8, vcc"
#define I_RNDNE(r) "v_rndne_f32 " #r ", " #r
#define I_ADDU(r) "v_add_u32 " #r ", " #r ", %8"
#define I_ADDSDWA(r) "v_add_f32_sdwa " #r ", " #r ", %8 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD"
#define I_ADDDPP(r) "v_add_f32_dpp " #r ", " #r ", %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf"
#define I_MOVDPP(r) "v_mov_b32_dpp " #r ", " #r " quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf"
#define I_ADDABS(r) "v_add_f32_e64 " #r ", |" #r "|, %8"
#define I_FMAMIX(r) "v_fma_mix_f32 " #r ", " #r ", %9, %8 op_sel_hi:[1,0,0]"
#define I_PKADDU16(r) "v_pk_add_u16 " #r ", " #r ", %8"
#define I_LSHLADD(r) "v_lshl_add_u32 " #r ", " #r ", 2, %8"
#define I_AND(r) "v_and_b32 " #r ", " #r ", %8"
#define I_BFE(r) "v_bfe_u32 " #r ", " #r ", 8, 8"
#define I_CVTPKI16(r) "v_cvt_pk_i16_i32 " #r ", " #r ", %8"
#define I_SATPK(r) "v_sat_pk_u8_i16 " #r ", " #r
#define I_PKFMA(r) "v_pk_fma_f32 " #r ", " #r ", %9, %8"
#define I_PKMUL(r) "v_pk_mul_f32 " #r ", " #r ", %9"

#define KERNEL(NAME, INSTR)                                                         \
    __global__ __launch_bounds__(256) void NAME(float *out, float a, float b, int iters) \
    {                                                                               \
        float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7; \
        for (int i = 0; i < iters; ++i) { CHAIN8(INSTR); CHAIN8(INSTR); }            \
        out[blockIdx.x * 256 + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;  \
    }
typedef float f2 __attribute__((ext_vector_type(2)));
#define KERNEL2(NAME, INSTR)                                                        \
    __global__ __launch_bounds__(256) void NAME(float *out, float fa, float fb, int iters) \
    {                                                                               \
        f2 a = fa, b = fb;                                                          \
        f2 x0 = (float)threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7; \
        for (int i = 0; i < iters; ++i) { CHAIN8(INSTR); CHAIN8(INSTR); }            \
        f2 s = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;                                \
        out[blockIdx.x * 256 + threadIdx.x] = s.x + s.y;                             \
    }

KERNEL(k_add, I_ADD) KERNEL(k_dot4, I_DOT4) KERNEL(k_dot2f, I_DOT2F) KERNEL(k_max, I_MAX) KERNEL(k_mov, I_MOV)
KERNEL(k_cndmask, I_CNDMASK) KERNEL(k_rndne, I_RNDNE) KERNEL(k_addu, I_ADDU) KERNEL(k_addsdwa, I_ADDSDWA)
KERNEL(k_adddpp, I_ADDDPP) KERNEL(k_movdpp, I_MOVDPP) KERNEL(k_addabs, I_ADDABS) KERNEL(k_fmamix, I_FMAMIX)
KERNEL(k_pkaddu16, I_PKADDU16) KERNEL(k_lshladd, I_LSHLADD) KERNEL(k_and, I_AND) KERNEL(k_bfe, I_BFE)
KERNEL(k_cvtpki16, I_CVTPKI16) KERNEL(k_satpk, I_SATPK) KERNEL2(k_pkfma, I_PKFMA) KERNEL2(k_pkmul, I_PKMUL)

typedef void (*kfn)(float *, float, float, int);

int main()
{
    float *d; (void)hipMalloc(&d, 256 * 2048 * 4);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    struct { const char *name; kfn fn; } ks[] = {
        {"v_add_f32", k_add}, {"v_dot4_u32_u8", k_dot4}, {"v_dot2_f32_f16", k_dot2f}, {"v_max_f32", k_max}, {"v_mov_b32", k_mov},
        {"v_cndmask_b32", k_cndmask}, {"v_rndne_f32", k_rndne}, {"v_add_u32", k_addu}, {"v_add_f32_sdwa (byte dst)", k_addsdwa},
        {"v_add_f32_dpp quad_perm", k_adddpp}, {"v_mov_b32_dpp quad_perm", k_movdpp}, {"v_add_f32 |abs| (VOP3)", k_addabs},
        {"v_fma_mix_f32", k_fmamix}, {"v_pk_add_u16", k_pkaddu16}, {"v_lshl_add_u32", k_lshladd}, {"v_and_b32", k_and},
        {"v_bfe_u32", k_bfe}, {"v_cvt_pk_i16_i32", k_cvtpki16}, {"v_sat_pk_u8_i16", k_satpk}, {"v_pk_fma_f32 (x2 elems)", k_pkfma},
        {"v_pk_mul_f32 (x2 elems)", k_pkmul}};
    const int iters = 10000;
    for (int blocks : {768, 2048}) {   // 3 and 8 waves per SIMD
        printf("-- %d waves per SIMD\n", blocks / 256);
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
            printf("%-28s %8.3f ms  %6.2f T instr-lanes/s\n", k.name, best, ops / best / 1e9);
        }
    }
    return 0;
}
