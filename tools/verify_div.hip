// verify_div.hip -- proof by exhaustion for the quantiser of the fused encode kernel.
//
// The reference divides by the modulated table with a true float division and then rounds half
// away from zero (encode.swift:225-240).  The fused kernel wants the 3-operation sequence
//     y0 = h * r;  e = fma(-y0, q, h);  y1 = fma(e, r, y0)      with r = RN(1 / q)
// (Markstein's correction step).  This program checks, on the GPU, that y1 is BIT-IDENTICAL to
// the correctly rounded quotient h / q for
//   * every divisor q = (r[k] r[h]) * (8 * Float(Q)) that an 8-bit quantisation table can produce
//     (Q = 1 .. 255, all 64 positions; duplicates removed), and
//   * EVERY non-negative float32 numerator below 2^17 (all 1.2e9 bit patterns; FDCT outputs of
//     8-bit samples are bounded by 64 * 255 * 1.4^2 * 8 < 2^17; negatives follow by symmetry).
// It also checks the rounding shortcut trunc(v + copysign(pred(0.5), v)) == round-half-away(v)
// for every such quotient actually produced.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <cmath>
#include <set>
#include <vector>

__global__ __launch_bounds__(256) void k_verify(const float *qs, int nq, uint32_t hbits_begin, uint32_t hbits_end, unsigned long long *bad_div,
                                                unsigned long long *bad_round)
{
    const uint32_t stride = gridDim.x * 256u;
    unsigned long long nbad = 0, nbadr = 0;
    for (int i = blockIdx.y; i < nq; i += gridDim.y) {
        const float q = qs[i];
        const float r = 1.0f / q;   // correctly rounded reciprocal (IEEE division)
        for (uint32_t b = hbits_begin + blockIdx.x * 256u + threadIdx.x; b < hbits_end; b += stride) {
            const float h = __uint_as_float(b);
            const float ref = h / q;
            const float y0 = h * r;
            const float e = __builtin_fmaf(-y0, q, h);
            const float y1 = __builtin_fmaf(e, r, y0);
            // (a) the quotient itself, bit for bit, wherever no intermediate can underflow
            if (b >= 0x20000000u || b == 0) nbad += __float_as_uint(ref) != __float_as_uint(y1);
            // (b) what the kernel stores: round-half-away of the reference quotient vs the
            //     shortcut applied to the fast quotient, for EVERY numerator
            const float want = roundf(ref);
            const float got = truncf(y1 + copysignf(0.49999997f, y1));
            nbadr += want != got;
        }
    }
    if (nbad) atomicAdd(bad_div, nbad);
    if (nbadr) atomicAdd(bad_round, nbadr);
}

int main(int argc, char **argv)
{
    const float rr[8] = {1.0f, 1.387039845f, 1.306562965f, 1.175875602f, 1.0f, 0.785694958f, 0.541196100f, 0.275899379f};
    std::set<uint32_t> uniq;
    for (int Q = 1; Q <= 255; ++Q)
        for (int h = 0; h < 8; ++h)
            for (int k = 0; k < 8; ++k) {
                const float hv = rr[k] * rr[h];
                const float row = 8.0f * (float)Q;
                const float q = hv * row;
                uint32_t bits; memcpy(&bits, &q, 4);
                uniq.insert(bits);
            }
    std::vector<float> qs;
    for (uint32_t b : uniq) { float f; memcpy(&f, &b, 4); qs.push_back(f); }
    int nq = (int)qs.size();
    if (argc > 1) nq = std::min(nq, atoi(argv[1]));   // quick mode: first N divisors
    const uint32_t hend = 0x48000000u;                // 2^17 as float bits: every pattern below it
    uint32_t hbegin = 0;
    if (argc > 2) hbegin = (uint32_t)strtoul(argv[2], nullptr, 16);
    printf("%d distinct divisors x numerator bit patterns [%08x, %08x) = %.3e quotients\n", nq, hbegin, hend, (double)nq * (hend - hbegin));

    float *d_q; unsigned long long *d_bad;
    (void)hipMalloc(&d_q, qs.size() * 4); (void)hipMalloc(&d_bad, 16);
    (void)hipMemcpy(d_q, qs.data(), qs.size() * 4, hipMemcpyHostToDevice);
    (void)hipMemset(d_bad, 0, 16);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k_verify, dim3(2048, 16), dim3(256), 0, 0, d_q, nq, hbegin, hend, d_bad, d_bad + 1);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    unsigned long long bad[2];
    (void)hipMemcpy(bad, d_bad, 16, hipMemcpyDeviceToHost);
    printf("quotient bit mismatches (numerators >= 2^-63 or 0): %llu   stored-integer mismatches (all numerators): %llu   (%.2f s)\n", bad[0], bad[1], ms / 1e3);
    return (bad[0] || bad[1]) ? 1 : 0;
}
