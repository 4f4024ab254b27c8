// fused_common.hpp -- device helpers shared by the fused decode kernels (kernels_fused.hip, kernels_quad.hip): division by
// launch invariants, LDS-DMA issue, the split arrive / wait on LDS counters, the per-phase cycle counters of the
// development build (-DJA_PHASE_PROFILE, tools/phase_profile.py).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <mutex>

namespace jpeg_amd {

constexpr int kThreads = 256;

// Division by a launch-invariant: q = mulhi(n, floor((2^32 - 1) / d)) is the quotient or one below it for every n < 2^32, one
// compare fixes it.  The strip walks divide a dozen times per trip (strip -> image, row, column; strips left); as true
// divisions that is ~400 mostly scalar, serial instructions at the head of every trip.
struct FastDiv {
    uint32_t d, m;
    __device__ __forceinline__ void set(uint32_t div) { d = div; m = 0xffffffffu / div; }
    __device__ __forceinline__ uint32_t div(uint32_t n, uint32_t &r) const
    {
        uint32_t q = __umulhi(n, m);
        r = n - q * d;
        if (r >= d) { ++q; r -= d; }
        return q;
    }
};

// A raw buffer resource (V#) for `bytes` bytes at `p`, built from wave-uniform values: base[47:0], stride 0, num_records = bytes,
// word 3 = 0x00020000 (gfx950: 32-bit data format, no swizzle).  buffer_load / buffer_store address base + soffset + voffset
// (+ the instruction's 12-bit offset) and range-check voffset (+ offset) against num_records - soffset: out-of-range loads
// return 0, out-of-range stores are dropped (tools/probe_buffer.hip, profiles/r05_probe_buffer.txt).
typedef int i32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ i32x4_t make_srd(const void *p, uint32_t bytes)
{
    const uint64_t a = reinterpret_cast<uint64_t>(p);
    i32x4_t r;
    r.x = (int)(uint32_t)a;
    r.y = (int)((uint32_t)(a >> 32) & 0xffffu);
    r.z = (int)bytes;
    r.w = 0x00020000;
    return r;
}

// One 16-byte-per-lane LDS-DMA: lane l's 16 B at `g` land at LDS byte address lds + 16 l.
// Issued from inline asm on purpose: hipcc cannot tell which LDS array a DMA targets, so
// after a __builtin_amdgcn_global_load_lds it makes the NEXT LDS read of any array wait with
// vmcnt(0) -- i.e. for the prefetch it was supposed to overlap.  Hidden from the compiler, the
// DMA is only waited for by the explicit s_waitcnt at the top of the next strip.  (Extra
// outstanding VM operations can only make the compiler's own counted waits longer, never
// shorter, because loads retire in order.)
__device__ __forceinline__ void lds_dma16(const void *g, uint32_t lds)
{
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off nt" ::"v"(g), "s"(lds) : "memory");
}
// Same with a scalar 64-bit base + per-lane 32-bit byte offset (no vector address arithmetic).
__device__ __forceinline__ void lds_dma16_s(uint64_t sbase, uint32_t voff, uint32_t lds)
{
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %0 nt" ::"s"(sbase), "v"(voff), "s"(lds) : "memory");
}
// The same two without the `nt` hint: for coefficients that neighbouring strips fetch again soon
// (the chroma blocks around a 4:2:0 strip) and should therefore stay in L2.
#ifdef JA_X_IN420_NT
#define JA_KEEP_HINT " nt"
#else
#define JA_KEEP_HINT ""
#endif
__device__ __forceinline__ void lds_dma16_keep(const void *g, uint32_t lds)
{
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" JA_KEEP_HINT ::"v"(g), "s"(lds) : "memory");
}
__device__ __forceinline__ void lds_dma16_s_keep(uint64_t sbase, uint32_t voff, uint32_t lds)
{
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %0" JA_KEEP_HINT ::"s"(sbase), "v"(voff), "s"(lds) : "memory");
}
// A RUN of N consecutive 1 KiB pieces (a contiguous source, a contiguous destination): the instruction's offset field moves
// the global source AND the LDS destination (tools/probe_dma_offset.hip), so the whole run needs one M0 write and one
// scalar base.  Piece j takes the per-lane offset v0 (even j) or v1 (odd j): the source-side swizzle alternates.
template <int N, bool NT>
__device__ __forceinline__ void lds_dma16_run(uint64_t sbase, uint32_t v0, uint32_t v1, uint32_t lds)
{
    static_assert(N == 2 || N == 4, "runs of 2 or 4 KiB");
    if constexpr (N == 2) {
        if constexpr (NT)
            asm volatile("s_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %0 nt\n\tglobal_load_lds_dwordx4 %2, %0 offset:1024 nt"
                         ::"s"(sbase), "v"(v0), "v"(v1), "s"(lds) : "memory");
        else
            asm volatile("s_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %0" JA_KEEP_HINT "\n\tglobal_load_lds_dwordx4 %2, %0 offset:1024" JA_KEEP_HINT
                         ::"s"(sbase), "v"(v0), "v"(v1), "s"(lds) : "memory");
    } else {
        if constexpr (NT)
            asm volatile("s_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %0 nt\n\tglobal_load_lds_dwordx4 %2, %0 offset:1024 nt\n\t"
                         "global_load_lds_dwordx4 %1, %0 offset:2048 nt\n\tglobal_load_lds_dwordx4 %2, %0 offset:3072 nt"
                         ::"s"(sbase), "v"(v0), "v"(v1), "s"(lds) : "memory");
        else
            asm volatile("s_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %0" JA_KEEP_HINT "\n\tglobal_load_lds_dwordx4 %2, %0 offset:1024" JA_KEEP_HINT "\n\t"
                         "global_load_lds_dwordx4 %1, %0 offset:2048" JA_KEEP_HINT "\n\tglobal_load_lds_dwordx4 %2, %0 offset:3072" JA_KEEP_HINT
                         ::"s"(sbase), "v"(v0), "v"(v1), "s"(lds) : "memory");
    }
}
// The same run through a buffer resource: source = srd.base + soff + per-lane offset (+ 1 KiB per piece), hardware range check
// against srd.num_records -- a lane whose 16 bytes lie outside the resource transfers ZEROS (tools/probe_buffer.hip T2), so
// block rows above / below a plane and runs that overhang its end need no clamped addresses.  `soff` must be a true,
// non-negative byte offset (or one that is out of range as an unsigned number).  s_nop 3: M0 is read one wait state after its
// write, and an SGPR operand the compiler happened to produce with v_readfirstlane five.
template <int N, bool NT>
__device__ __forceinline__ void lds_dma16_brun(i32x4_t srd, uint32_t soff, uint32_t v0, uint32_t v1, uint32_t lds)
{
    static_assert(N == 1 || N == 2 || N == 4, "runs of 1, 2 or 4 KiB");
#define JA_BL(v, off, hint) "buffer_load_dwordx4 " v ", %0, %1 offen" off hint " lds"
    if constexpr (N == 1) {
        if constexpr (NT) asm volatile("s_mov_b32 m0, %4\n\ts_nop 3\n\t" JA_BL("%2", "", " nt") ::"s"(srd), "s"(soff), "v"(v0), "v"(v1), "s"(lds) : "memory");
        else asm volatile("s_mov_b32 m0, %4\n\ts_nop 3\n\t" JA_BL("%2", "", JA_KEEP_HINT) ::"s"(srd), "s"(soff), "v"(v0), "v"(v1), "s"(lds) : "memory");
    } else if constexpr (N == 2) {
        if constexpr (NT) asm volatile("s_mov_b32 m0, %4\n\ts_nop 3\n\t" JA_BL("%2", "", " nt") "\n\t" JA_BL("%3", " offset:1024", " nt")
                                       ::"s"(srd), "s"(soff), "v"(v0), "v"(v1), "s"(lds) : "memory");
        else asm volatile("s_mov_b32 m0, %4\n\ts_nop 3\n\t" JA_BL("%2", "", JA_KEEP_HINT) "\n\t" JA_BL("%3", " offset:1024", JA_KEEP_HINT)
                          ::"s"(srd), "s"(soff), "v"(v0), "v"(v1), "s"(lds) : "memory");
    } else {
        if constexpr (NT) asm volatile("s_mov_b32 m0, %4\n\ts_nop 3\n\t" JA_BL("%2", "", " nt") "\n\t" JA_BL("%3", " offset:1024", " nt") "\n\t"
                                       JA_BL("%2", " offset:2048", " nt") "\n\t" JA_BL("%3", " offset:3072", " nt")
                                       ::"s"(srd), "s"(soff), "v"(v0), "v"(v1), "s"(lds) : "memory");
        else asm volatile("s_mov_b32 m0, %4\n\ts_nop 3\n\t" JA_BL("%2", "", JA_KEEP_HINT) "\n\t" JA_BL("%3", " offset:1024", JA_KEEP_HINT) "\n\t"
                          JA_BL("%2", " offset:2048", JA_KEEP_HINT) "\n\t" JA_BL("%3", " offset:3072", JA_KEEP_HINT)
                          ::"s"(srd), "s"(soff), "v"(v0), "v"(v1), "s"(lds) : "memory");
    }
#undef JA_BL
}
// 4 bytes per lane: lane l's dword lands at LDS byte address lds + 4 l.
__device__ __forceinline__ void lds_dma4_s(uint64_t sbase, uint32_t voff, uint32_t lds)
{
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, %0" ::"s"(sbase), "v"(voff), "s"(lds) : "memory");
}

// Resident workgroups of a persistent kernel on the CURRENT device: workgroups per CU (what the occupancy query says for this
// instantiation) x CUs.  One cache per KERNEL -- the kernel's address is the template argument: all instantiations of a
// kernel template share one function TYPE, so a cache keyed by type would hand the first instantiation's answer to all of
// them -- and per device, filled once under std::call_once: contexts of several devices, and first calls from several host
// threads, see their own device's value.
constexpr int kMaxDevices = 64;
template <auto Kernel>
inline int resident_workgroups_of(int fallback_per_cu)
{
    struct PerDevice { std::once_flag once; int value = 0; };
    static PerDevice cache[kMaxDevices];   // one array per kernel instantiation
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0) dev = 0;
    auto query = [&]() {
        int per_cu = 0, cus = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, Kernel, kThreads, 0) != hipSuccess || per_cu < 1) per_cu = fallback_per_cu;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1) cus = 256;
        return per_cu * cus;
    };
    if (dev >= kMaxDevices) return query();
    std::call_once(cache[dev].once, [&]() { cache[dev].value = query(); });
    return cache[dev].value;
}

// LDS byte address of a __shared__ object (low 32 bits of its flat address), wave-uniform
template <typename T>
__device__ __forceinline__ uint32_t lds_address(T *p)
{
    return (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(uintptr_t)(__attribute__((address_space(3))) T *)p);
}

// Split arrive / wait on an LDS counter (the 4:2:0 stack walk).  The LDS executes one wave's operations in issue order, so a
// ds_add issued behind the wave's tile writes (or its last tile reads) is performed behind them: no s_waitcnt at the arrive.
// The waiting side polls with plain LDS reads; what it reads from the tile after the poll has succeeded is issued, and
// therefore performed, after the read that saw the counter.  Relaxed atomics + compiler barriers on purpose: a release /
// acquire at workgroup scope would make hipcc wait with vmcnt(0) -- for the coefficient DMA in flight and for the pixel stores.
// One lane adds, selected by narrowing EXEC around the ds_add (the compiler's form of "if (lane == 0) atomicAdd" is a dozen
// instructions: compare, save EXEC, count the active lanes, multiply, restore).  EXEC is saved and restored, not assumed:
// the helper is correct from a divergent region as long as lane 0 is active there (every call site runs with all 64 lanes).
__device__ __forceinline__ void lds_arrive(uint32_t *counter)
{
    const uint32_t addr = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint32_t *)counter;
#ifdef JA_DEBUG_ASSERTS   // the precondition, checked in debug builds: lane 0 is active here
    if (!(__builtin_amdgcn_read_exec() & 1ull)) __builtin_trap();
#endif
    uint64_t saved;
    asm volatile("s_mov_b64 %0, exec\n\ts_mov_b64 exec, 1\n\tds_add_u32 %1, %2\n\ts_mov_b64 exec, %0" : "=&s"(saved) : "v"(addr), "v"(1u) : "memory");
}
__device__ __forceinline__ uint32_t lds_peek(uint32_t *counter)
{
    return __hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ void lds_wait_ge(uint32_t *counter, uint32_t target)
{
#ifdef JA_X_NOSYNC   // experiment (wrong pixels): what do the waits of the stack walk cost?
    return;
#endif
    asm volatile("" ::: "memory");
    while ((uint32_t)__builtin_amdgcn_readfirstlane((int)lds_peek(counter)) < target) __builtin_amdgcn_s_sleep(1);
    asm volatile("" ::: "memory");
}
// the same when the counter was already read a while ago (`seen`, any lane's copy): the common case costs no LDS round trip
__device__ __forceinline__ void lds_wait_ge_seen(uint32_t *counter, uint32_t target, uint32_t seen)
{
#ifdef JA_X_NOSYNC
    return;
#endif
    if ((uint32_t)__builtin_amdgcn_readfirstlane((int)seen) < target) lds_wait_ge(counter, target);
    asm volatile("" ::: "memory");
}

// ---- clamp [0, 255] + truncate + pack, the reference's float -> byte conversion (decode.swift:4121-4122, jpeg.swift:343-354)
// in ONE instruction per sample.  v_cvt_pk_u8_f32 saturates and rounds in the wave's f32 rounding mode
// (tools/probe_cvt_round.hip, profiles/r03_probe_cvt_round.txt): under round-toward-zero it truncates, where the default
// mode needs a v_floor_f32 in front of it (two half-rate instructions per sample).  The mode is switched and restored
// inside one asm statement, so nothing the compiler schedules can land between the two s_setreg; the f32 instructions
// right before and right after the statement keep round-to-nearest (tools/probe_cvt_round2.hip).
// c: 4 n floats, d: n dwords (byte i of d[k] = c[4 k + i]).
// Precondition of every trunc_* helper: the wave runs in the default f32 rounding mode (round to nearest even, MODE[1:0] =
// 0) -- the statement restores THAT, not a saved value (s_getreg + s_setreg_b32 would cost two more scalar slots per
// pack; no kernel of this library ever leaves another mode set).
#ifdef JA_X_NOSETREG   // experiment (wrong pixels): the packs without the two mode switches
#define JA_RTZ_ON ""
#define JA_RTZ_OFF ""
#else
#define JA_RTZ_ON "s_setreg_imm32_b32 hwreg(HW_REG_MODE, 0, 2), 3\n\t"
#define JA_RTZ_OFF "s_setreg_imm32_b32 hwreg(HW_REG_MODE, 0, 2), 0"
#endif
__device__ __forceinline__ void trunc_pack8(const float *c, uint32_t *d)
{
    asm volatile(JA_RTZ_ON
                 "v_cvt_pk_u8_f32 %0, %2, 0, 0\n\t"
                 "v_cvt_pk_u8_f32 %0, %3, 1, %0\n\t"
                 "v_cvt_pk_u8_f32 %0, %4, 2, %0\n\t"
                 "v_cvt_pk_u8_f32 %0, %5, 3, %0\n\t"
                 "v_cvt_pk_u8_f32 %1, %6, 0, 0\n\t"
                 "v_cvt_pk_u8_f32 %1, %7, 1, %1\n\t"
                 "v_cvt_pk_u8_f32 %1, %8, 2, %1\n\t"
                 "v_cvt_pk_u8_f32 %1, %9, 3, %1\n\t"
                 JA_RTZ_OFF
                 : "=&v"(d[0]), "=&v"(d[1])
                 : "v"(c[0]), "v"(c[1]), "v"(c[2]), "v"(c[3]), "v"(c[4]), "v"(c[5]), "v"(c[6]), "v"(c[7]));
}
__device__ __forceinline__ void trunc_pack16(const float *c, uint32_t *d)
{
    asm volatile(JA_RTZ_ON
                 "v_cvt_pk_u8_f32 %0, %4, 0, 0\n\t"
                 "v_cvt_pk_u8_f32 %0, %5, 1, %0\n\t"
                 "v_cvt_pk_u8_f32 %0, %6, 2, %0\n\t"
                 "v_cvt_pk_u8_f32 %0, %7, 3, %0\n\t"
                 "v_cvt_pk_u8_f32 %1, %8, 0, 0\n\t"
                 "v_cvt_pk_u8_f32 %1, %9, 1, %1\n\t"
                 "v_cvt_pk_u8_f32 %1, %10, 2, %1\n\t"
                 "v_cvt_pk_u8_f32 %1, %11, 3, %1\n\t"
                 "v_cvt_pk_u8_f32 %2, %12, 0, 0\n\t"
                 "v_cvt_pk_u8_f32 %2, %13, 1, %2\n\t"
                 "v_cvt_pk_u8_f32 %2, %14, 2, %2\n\t"
                 "v_cvt_pk_u8_f32 %2, %15, 3, %2\n\t"
                 "v_cvt_pk_u8_f32 %3, %16, 0, 0\n\t"
                 "v_cvt_pk_u8_f32 %3, %17, 1, %3\n\t"
                 "v_cvt_pk_u8_f32 %3, %18, 2, %3\n\t"
                 "v_cvt_pk_u8_f32 %3, %19, 3, %3\n\t"
                 JA_RTZ_OFF
                 : "=&v"(d[0]), "=&v"(d[1]), "=&v"(d[2]), "=&v"(d[3])
                 : "v"(c[0]), "v"(c[1]), "v"(c[2]), "v"(c[3]), "v"(c[4]), "v"(c[5]), "v"(c[6]), "v"(c[7]), "v"(c[8]), "v"(c[9]), "v"(c[10]), "v"(c[11]), "v"(c[12]), "v"(c[13]), "v"(c[14]), "v"(c[15]));
}
__device__ __forceinline__ void trunc_pack24(const float *c, uint32_t *d)
{
    asm volatile(JA_RTZ_ON
                 "v_cvt_pk_u8_f32 %0, %6, 0, 0\n\t"
                 "v_cvt_pk_u8_f32 %0, %7, 1, %0\n\t"
                 "v_cvt_pk_u8_f32 %0, %8, 2, %0\n\t"
                 "v_cvt_pk_u8_f32 %0, %9, 3, %0\n\t"
                 "v_cvt_pk_u8_f32 %1, %10, 0, 0\n\t"
                 "v_cvt_pk_u8_f32 %1, %11, 1, %1\n\t"
                 "v_cvt_pk_u8_f32 %1, %12, 2, %1\n\t"
                 "v_cvt_pk_u8_f32 %1, %13, 3, %1\n\t"
                 "v_cvt_pk_u8_f32 %2, %14, 0, 0\n\t"
                 "v_cvt_pk_u8_f32 %2, %15, 1, %2\n\t"
                 "v_cvt_pk_u8_f32 %2, %16, 2, %2\n\t"
                 "v_cvt_pk_u8_f32 %2, %17, 3, %2\n\t"
                 "v_cvt_pk_u8_f32 %3, %18, 0, 0\n\t"
                 "v_cvt_pk_u8_f32 %3, %19, 1, %3\n\t"
                 "v_cvt_pk_u8_f32 %3, %20, 2, %3\n\t"
                 "v_cvt_pk_u8_f32 %3, %21, 3, %3\n\t"
                 "v_cvt_pk_u8_f32 %4, %22, 0, 0\n\t"
                 "v_cvt_pk_u8_f32 %4, %23, 1, %4\n\t"
                 "v_cvt_pk_u8_f32 %4, %24, 2, %4\n\t"
                 "v_cvt_pk_u8_f32 %4, %25, 3, %4\n\t"
                 "v_cvt_pk_u8_f32 %5, %26, 0, 0\n\t"
                 "v_cvt_pk_u8_f32 %5, %27, 1, %5\n\t"
                 "v_cvt_pk_u8_f32 %5, %28, 2, %5\n\t"
                 "v_cvt_pk_u8_f32 %5, %29, 3, %5\n\t"
                 JA_RTZ_OFF
                 : "=&v"(d[0]), "=&v"(d[1]), "=&v"(d[2]), "=&v"(d[3]), "=&v"(d[4]), "=&v"(d[5])
                 : "v"(c[0]), "v"(c[1]), "v"(c[2]), "v"(c[3]), "v"(c[4]), "v"(c[5]), "v"(c[6]), "v"(c[7]), "v"(c[8]), "v"(c[9]), "v"(c[10]), "v"(c[11]), "v"(c[12]), "v"(c[13]), "v"(c[14]), "v"(c[15]), "v"(c[16]), "v"(c[17]), "v"(c[18]), "v"(c[19]), "v"(c[20]), "v"(c[21]), "v"(c[22]), "v"(c[23]));
}
// eight samples, each into byte 0 of its own dword (the other bytes 0)
__device__ __forceinline__ void trunc_bytes8(const float *c, uint32_t *d)
{
    asm volatile(JA_RTZ_ON
                 "v_cvt_pk_u8_f32 %0, %8, 0, 0\n\t"
                 "v_cvt_pk_u8_f32 %1, %9, 0, 0\n\t"
                 "v_cvt_pk_u8_f32 %2, %10, 0, 0\n\t"
                 "v_cvt_pk_u8_f32 %3, %11, 0, 0\n\t"
                 "v_cvt_pk_u8_f32 %4, %12, 0, 0\n\t"
                 "v_cvt_pk_u8_f32 %5, %13, 0, 0\n\t"
                 "v_cvt_pk_u8_f32 %6, %14, 0, 0\n\t"
                 "v_cvt_pk_u8_f32 %7, %15, 0, 0\n\t"
                 JA_RTZ_OFF
                 : "=&v"(d[0]), "=&v"(d[1]), "=&v"(d[2]), "=&v"(d[3]), "=&v"(d[4]), "=&v"(d[5]), "=&v"(d[6]), "=&v"(d[7])
                 : "v"(c[0]), "v"(c[1]), "v"(c[2]), "v"(c[3]), "v"(c[4]), "v"(c[5]), "v"(c[6]), "v"(c[7]));
}

#ifdef JA_PHASE_PROFILE
// development aid (tools/phase_profile.py): wall cycles each wave spends per phase of a strip
constexpr int kPhaseSlots = 16;   // 0..13 phases, 14 the wave's life in shader cycles, 15 in ticks of the 100 MHz counter
// each translation unit that profiles holds its own copy (no relocatable device code): JA_PHASE_STORAGE at namespace scope
#define JA_PHASE_STORAGE                                                                                \
    __device__ unsigned long long g_phase_cycles[4096 * kPhaseSlots];                                   \
    __device__ unsigned long long g_wave_info[4096 * 4];   /* start tick, end tick (100 MHz counter), HW_ID, XCC_ID */
#define JA_PHASE_DECL                                                                                   \
    unsigned long long phase_acc[kPhaseSlots] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};      \
    unsigned long long t_prev = __builtin_readcyclecounter();                                           \
    const unsigned long long t_first = t_prev, r_first = __builtin_amdgcn_s_memrealtime();
#define JA_PHASE(i)                                                       \
    {                                                                     \
        const unsigned long long t_now = __builtin_readcyclecounter();    \
        phase_acc[i] += t_now - t_prev;                                   \
        t_prev = t_now;                                                   \
    }
#define JA_PHASE_FLUSH(slot, lane)                                                                       \
    {                                                                                                    \
        phase_acc[14] = __builtin_readcyclecounter() - t_first;                                          \
        phase_acc[15] = __builtin_amdgcn_s_memrealtime() - r_first;                                      \
        if ((lane) == 0 && (slot) < 4096) {                                                              \
            for (int i_ = 0; i_ < kPhaseSlots; ++i_) g_phase_cycles[(slot) * kPhaseSlots + i_] = phase_acc[i_]; \
            unsigned long long *wi_ = g_wave_info + (slot) * 4;                                          \
            wi_[0] = r_first; wi_[1] = r_first + phase_acc[15];                                          \
            unsigned hw_, xcc_;                                                                          \
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw_));                            \
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc_));                          \
            wi_[2] = hw_; wi_[3] = xcc_ & 15u;                                                           \
        }                                                                                                \
    }
#else
#define JA_PHASE_STORAGE
#define JA_PHASE_DECL
#define JA_PHASE(i)
#define JA_PHASE_FLUSH(slot, lane)
#endif

}  // namespace jpeg_amd
