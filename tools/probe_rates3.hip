// probe_rates3.hip -- round 6: issue cost of the opcodes the round-6 encode kernel considers (same harness as probe_valu_classes.hip) -- round 5: the cost table behind `roofline.valu` (tools/valu_roofline.py).
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
KERNEL(0, E8D("v_add_f32_e32 ", ", %8, ", ""))
KERNEL(1, E8D("v_cvt_pk_u8_f32 ", ", %8, 1, ", ""))
KERNEL(2, E8D("v_fma_mix_f32 ", ", ", ", %8, %9 op_sel_hi:[1,0,0]"))
KERNEL(3, E8D("v_fma_mix_f32 ", ", %8, ", ", %9 op_sel:[0,1,0] op_sel_hi:[0,1,0]"))
KERNEL(4, E8D("v_cvt_i32_f32_sdwa ", ", ", " dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD"))
KERNEL(5, E8D("v_pack_b32_f16 ", ", ", ", %8"))
KERNEL(6, E8D("v_cvt_f32_f16_e32 ", ", ", ""))
KERNEL(7, E8D("v_cvt_pkrtz_f16_f32 ", ", ", ", %8"))
KERNEL(8, E8D("v_pk_add_u16 ", ", ", ", %8"))
KERNEL(9, E8D("v_pk_mad_u16 ", ", ", ", %8, %9"))
KERNEL(10, E8D("v_cvt_rpi_i32_f32_e32 ", ", ", ""))
KERNEL(11, E8D("v_alignbit_b32 ", ", ", ", %8, 16"))
KERNEL(12, E8D("v_bitop3_b32 ", ", ", ", %8, %9 bitop3:0xca"))
KERNEL(13, E8D("v_mul_f32_e64 ", ", |", "|, %8"))
KERNEL(14, E8D("v_cvt_f32_f16_sdwa ", ", ", " dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1"))
KERNEL(15, E8D("v_trunc_f32_e32 ", ", ", ""))
KERNEL(16, E8D("v_fract_f32_e32 ", ", ", ""))
KERNEL(17, E8D("v_sub_u32_e32 ", ", %8, ", ""))
KERNEL(18, E8D("v_lshrrev_b32_e32 ", ", 1, ", ""))
KERNEL(19, E8D("v_mad_i32_i24 ", ", ", ", %8, %9"))
constexpr int kOps = 20;
static const char *kNames[kOps] = {"v_add_f32", "v_cvt_pk_u8_f32", "v_fma_mix_f32 (f16 lo src0)", "v_fma_mix_f32 (f16 hi src1)", "v_cvt_i32_f32_sdwa W1 keep", "v_pack_b32_f16", "v_cvt_f32_f16", "v_cvt_pkrtz_f16_f32", "v_pk_add_u16", "v_pk_mad_u16", "v_cvt_rpi_i32_f32", "v_alignbit_b32", "v_bitop3_b32 0xca", "v_mul_f32 |abs| (e64)", "v_cvt_f32_f16_sdwa W1", "v_trunc_f32", "v_fract_f32", "v_sub_u32", "v_lshrrev_b32", "v_mad_i32_i24"};

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
    printf("probe_rates3: %d CUs, %d iterations x %d instructions per wave.  Per opcode and waves per SIMD (8 / 4 / 3):\n"
           "  ticks = s_memtime ticks per instruction and SIMD (mean wave time / (waves per SIMD x instructions));\n"
           "  ns    = the same from the wall clock (HIP events: kernel time / instructions per SIMD);\n"
           "  MHz   = s_memtime ticks per microsecond of s_memrealtime (100 MHz): what the tick counter ran at\n", cus, iters, 8 * kUnroll);
    printf("%3s %-28s | %7s %7s %6s | %7s %7s %6s | %7s %7s %6s\n", "id", "opcode", "tick@8w", "ns@8w", "MHz", "tick@4w", "ns@4w", "MHz", "tick@3w", "ns@3w", "MHz");
    for (int id = 0; id < kOps; ++id) {
        printf("%3d %-28s", id, kNames[id]);
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
