// probe_valu_classes.hip -- round 5: the cost table behind `roofline.valu` (tools/valu_roofline.py).
// For every VALU opcode the fused kernels execute in numbers: (a) issue cycles per instruction and SIMD with 8 and with 3
// (and 4) resident waves per SIMD, measured in SHADER cycles (s_memtime) -- clock-independent; (b) run under
//   rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32
//             SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 -- ./tools/probe_valu_classes
// which SQ counter class the opcode is tallied in (one kernel per opcode: k_op<ID>; the table ID -> opcode is printed).
// build: hipcc --offload-arch=gfx950 -O2 tools/probe_valu_classes.hip -o tools/probe_valu_classes
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <algorithm>

constexpr int kUnroll = 4;      // asm statements per loop trip, each 8 instructions (one per chain)

template <int ID> struct Op;
// eight copies of one instruction, chain k in register %k
#define E8(a, b) a "%0" b "\n\t" a "%1" b "\n\t" a "%2" b "\n\t" a "%3" b "\n\t" a "%4" b "\n\t" a "%5" b "\n\t" a "%6" b "\n\t" a "%7" b
#define E8D(a, m, b) a "%0" m "%0" b "\n\t" a "%1" m "%1" b "\n\t" a "%2" m "%2" b "\n\t" a "%3" m "%3" b "\n\t" a "%4" m "%4" b "\n\t" a "%5" m "%5" b "\n\t" a "%6" m "%6" b "\n\t" a "%7" m "%7" b

#define KERNEL(id, text)                                                                                          \
    template <> struct Op<id> {                                                                                   \
        static __device__ __forceinline__ void run(float (&c)[8], float s1, float s2)                             \
        {                                                                                                         \
            asm volatile(text : "+v"(c[0]), "+v"(c[1]), "+v"(c[2]), "+v"(c[3]), "+v"(c[4]), "+v"(c[5]), "+v"(c[6]), "+v"(c[7]) \
                         : "v"(s1), "v"(s2) : "vcc");                                                             \
        }                                                                                                         \
    };
KERNEL(0,  E8D("v_add_f32_e32 ", ", %8, ", ""))
KERNEL(1,  E8D("v_sub_f32_e32 ", ", ", ", %8"))
KERNEL(2,  E8D("v_mul_f32_e32 ", ", %8, ", ""))
KERNEL(3,  E8("v_fmac_f32_e32 ", ", %8, %9"))
KERNEL(4,  E8D("v_fmamk_f32 ", ", ", ", 0x3d800000, %8"))
KERNEL(5,  E8D("v_fma_f32 ", ", ", ", %8, %9"))
KERNEL(6,  E8D("v_cvt_pk_u8_f32 ", ", %8, 1, ", ""))
KERNEL(7,  E8D("v_cvt_f32_i32_sdwa ", ", sext(", ") dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1"))
KERNEL(8,  E8D("v_cvt_f32_ubyte1_e32 ", ", ", ""))
KERNEL(9,  E8D("v_perm_b32 ", ", ", ", %8, %9"))
KERNEL(10, E8D("v_cndmask_b32_e32 ", ", %8, ", ", vcc"))
KERNEL(11, E8("v_mov_b32_e32 ", ", %8"))
KERNEL(12, E8D("v_add_u32_e32 ", ", %8, ", ""))
KERNEL(13, E8D("v_and_b32_e32 ", ", %8, ", ""))
KERNEL(14, E8D("v_xor_b32_e32 ", ", %8, ", ""))
KERNEL(15, E8D("v_lshlrev_b32_e32 ", ", 1, ", ""))
KERNEL(16, E8D("v_lshrrev_b32_e32 ", ", 1, ", ""))
KERNEL(17, E8D("v_lshl_add_u32 ", ", ", ", 1, %8"))
KERNEL(18, E8D("v_add3_u32 ", ", ", ", %8, %9"))
KERNEL(19, E8D("v_mul_lo_u32 ", ", ", ", %8"))
KERNEL(20, E8D("v_floor_f32_e32 ", ", ", ""))
KERNEL(21, E8D("v_max_f32_e32 ", ", %8, ", ""))
KERNEL(22, E8D("v_med3_f32 ", ", ", ", %8, %9"))
KERNEL(23, E8D("v_cvt_i32_f32_e32 ", ", ", ""))
KERNEL(24, E8D("v_cvt_pk_i16_i32 ", ", ", ", %8"))
KERNEL(25, E8D("v_rcp_f32_e32 ", ", ", ""))
KERNEL(26, E8D("v_and_or_b32 ", ", ", ", %8, %9"))
KERNEL(27, E8("v_cmp_gt_i32_e32 vcc, %8, ", ""))
KERNEL(28, E8D("v_bitop3_b32 ", ", ", ", %8, %9 bitop3:0x96"))
KERNEL(29, E8D("v_cvt_f32_i32_e32 ", ", ", ""))
KERNEL(30, E8D("v_or_b32_e32 ", ", %8, ", ""))
KERNEL(31, E8D("v_dot4_u32_u8 ", ", ", ", %8, %9"))
KERNEL(32, E8D("v_add_f32_dpp ", ", ", ", %8 row_shr:1 row_mask:0xf bank_mask:0xf"))
KERNEL(33, E8D("v_ashrrev_i32_e32 ", ", 1, ", ""))
KERNEL(34, E8D("v_min_i32_e32 ", ", %8, ", ""))
KERNEL(35, E8D("v_bfi_b32 ", ", %8, ", ", %9"))
KERNEL(36, E8D("v_cvt_f32_ubyte0_e32 ", ", ", ""))
KERNEL(37, E8D("v_rndne_f32_e32 ", ", ", ""))
KERNEL(38, E8D("v_mul_u32_u24_e32 ", ", %8, ", ""))
KERNEL(39, E8D("v_mad_u32_u24 ", ", ", ", %8, %9"))
KERNEL(40, E8D("v_min_f32_e32 ", ", %8, ", ""))
KERNEL(41, E8D("v_lshl_or_b32 ", ", ", ", 1, %8"))
constexpr int kOps = 42;
static const char *kNames[kOps] = {"v_add_f32", "v_sub_f32", "v_mul_f32", "v_fmac_f32", "v_fmamk_f32", "v_fma_f32", "v_cvt_pk_u8_f32", "v_cvt_f32_i32_sdwa",
    "v_cvt_f32_ubyte1", "v_perm_b32", "v_cndmask_b32", "v_mov_b32", "v_add_u32", "v_and_b32", "v_xor_b32", "v_lshlrev_b32", "v_lshrrev_b32",
    "v_lshl_add_u32", "v_add3_u32", "v_mul_lo_u32", "v_floor_f32", "v_max_f32", "v_med3_f32", "v_cvt_i32_f32", "v_cvt_pk_i16_i32", "v_rcp_f32",
    "v_and_or_b32", "v_cmp_gt_i32", "v_bitop3_b32", "v_cvt_f32_i32", "v_or_b32", "v_dot4_u32_u8", "v_add_f32_dpp", "v_ashrrev_i32", "v_min_i32",
    "v_bfi_b32", "v_cvt_f32_ubyte0", "v_rndne_f32", "v_mul_u32_u24", "v_mad_u32_u24", "v_min_f32", "v_lshl_or_b32"};

template <int ID>
__global__ __launch_bounds__(256) void k_op(float *out, unsigned long long *cyc, unsigned long long *real, int iters)
{
    float c[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) c[i] = 1.0f + 0.001f * (float)(threadIdx.x + i);
    const float s1 = 1.0000001f, s2 = 0.5f;
    __syncthreads();
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < kUnroll; ++u) Op<ID>::run(c, s1, s2);
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    float acc = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) acc += c[i];
    if (acc == 123.456f) out[threadIdx.x] = acc;   // keep the chains alive
    if ((threadIdx.x & 63) == 0) { cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0; real[blockIdx.x * 4 + (threadIdx.x >> 6)] = r1 - r0; }
}

typedef void (*kern_t)(float *, unsigned long long *, unsigned long long *, int);
template <int... I> static void fill(kern_t *t, std::integer_sequence<int, I...>) { ((t[I] = k_op<I>), ...); }

int main(int argc, char **argv)
{
    const int iters = argc > 1 ? atoi(argv[1]) : 2000;
    kern_t table[kOps];
    fill(table, std::make_integer_sequence<int, kOps>{});
    int cus = 256;
    hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    float *out; unsigned long long *cyc, *real;
    hipMalloc(&out, 4096); hipMalloc(&cyc, 8 * 65536); hipMalloc(&real, 8 * 65536);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    printf("probe_valu_classes: %d CUs, %d iterations x %d instructions per wave.  Per opcode and waves per SIMD (8 / 4 / 3):\n"
           "  ticks = s_memtime ticks per instruction and SIMD (mean wave time / (waves per SIMD x instructions));\n"
           "  ns    = the same from the wall clock (HIP events: kernel time / instructions per SIMD);\n"
           "  MHz   = s_memtime ticks per microsecond of s_memrealtime (100 MHz): what the tick counter ran at\n", cus, iters, 8 * kUnroll);
    printf("%3s %-22s | %7s %7s %6s | %7s %7s %6s | %7s %7s %6s\n", "id", "opcode", "tick@8w", "ns@8w", "MHz", "tick@4w", "ns@4w", "MHz", "tick@3w", "ns@3w", "MHz");
    for (int id = 0; id < kOps; ++id) {
        printf("%3d %-22s", id, kNames[id]);
        for (int wps : {8, 4, 3}) {   // waves per SIMD: blocks of 256 work-items = one wave per SIMD each; wps blocks per CU
            const int blocks = cus * wps;
            hipLaunchKernelGGL(table[id], dim3(blocks), dim3(256), 0, 0, out, cyc, real, 50);   // warm
            hipEventRecord(e0, 0);
            hipLaunchKernelGGL(table[id], dim3(blocks), dim3(256), 0, 0, out, cyc, real, iters);
            hipEventRecord(e1, 0);
            hipDeviceSynchronize();
            float ms = 0; hipEventElapsedTime(&ms, e0, e1);
            std::vector<unsigned long long> h(blocks * 4), hr(blocks * 4);
            hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
            hipMemcpy(hr.data(), real, hr.size() * 8, hipMemcpyDeviceToHost);
            double mean = 0, meanr = 0;
            for (auto v : h) mean += (double)v;
            for (auto v : hr) meanr += (double)v;
            mean /= (double)h.size(); meanr /= (double)hr.size();
            const double n = (double)wps * iters * 8 * kUnroll;   // instructions per SIMD
            printf(" | %7.3f %7.3f %6.0f", mean / n, ms * 1e6 / n, mean / (meanr / 100.0));
        }
        printf("\n");
    }
    return 0;
}
