// verify_div16.hip -- round 6: the 3-operation quantiser division for EVERY JPEG.Format (k_generic_encode).
//
// tools/verify_div.hip proved  y0 = h * r;  e = fma(-y0, q, h);  y1 = fma(e, r, y0),  r = RN(1 / q)  bit-identical to the IEEE
// quotient h / q for the divisors of 8-bit tables and numerators below 2^17 (k_encode_fused).  The generic encode kernel serves
// precisions up to 16 bits and 16-bit tables: this program checks the same identity -- and the stored integer,
// trunc(y1 + copysign(pred(1/2), y1)) against roundf(h / q) -- for
//   * every divisor q = (r[k] r[h]) * (8 * Float(Q)), Q = Q0 .. Q1 - 1 of 1 .. 65535, all 64 positions (duplicates removed), and
//   * EVERY non-negative float32 numerator below 2^25 (FDCT outputs of 16-bit samples: 64 * 65535 * 1.39^2 * ... < 2^25).
// usage: verify_div16 Q0 Q1 [stride]   (stride: every stride-th Q, for a quick sample)
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <set>
#include <vector>

__global__ __launch_bounds__(256) void k_verify(const float *qs, int nq, uint32_t hbits_begin, uint32_t hbits_end, unsigned long long *bad_div,
                                                unsigned long long *bad_round, uint32_t *first_bad)
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
            const bool bd = (b >= 0x20000000u || b == 0) && __float_as_uint(ref) != __float_as_uint(y1);
            const float want = roundf(ref);
            const float got = truncf(y1 + copysignf(0.49999997f, y1));
            const bool br = want != got;
            nbad += bd; nbadr += br;
            if (br) { first_bad[0] = __float_as_uint(q); first_bad[1] = b; }
        }
    }
    if (nbad) atomicAdd(bad_div, nbad);
    if (nbadr) atomicAdd(bad_round, nbadr);
}

int main(int argc, char **argv)
{
    const int Q0 = argc > 1 ? atoi(argv[1]) : 1, Q1 = argc > 2 ? atoi(argv[2]) : 65536, step = argc > 3 ? atoi(argv[3]) : 1;
    const float rr[8] = {1.0f, 1.387039845f, 1.306562965f, 1.175875602f, 1.0f, 0.785694958f, 0.541196100f, 0.275899379f};
    std::set<uint32_t> uniq;
    for (int Q = Q0; Q < Q1; Q += step)
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
    const int nq = (int)qs.size();
    const uint32_t hend = 0x4C000000u;                // 2^25 as float bits: every pattern below it
    printf("Q = %d .. %d step %d: %d distinct divisors x numerator bit patterns [0, %08x) = %.3e quotients\n", Q0, Q1 - 1, step, nq, hend, (double)nq * hend);
    float *d_q; unsigned long long *d_bad; uint32_t *d_first;
    (void)hipMalloc(&d_q, qs.size() * 4); (void)hipMalloc(&d_bad, 16); (void)hipMalloc(&d_first, 8);
    (void)hipMemcpy(d_q, qs.data(), qs.size() * 4, hipMemcpyHostToDevice);
    (void)hipMemset(d_bad, 0, 16); (void)hipMemset(d_first, 0, 8);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0);
    // in slices of 4096 divisors: a launch of a few seconds each
    for (int at = 0; at < nq; at += 4096)
        hipLaunchKernelGGL(k_verify, dim3(2048, 16), dim3(256), 0, 0, d_q + at, std::min(4096, nq - at), 0u, hend, d_bad, d_bad + 1, d_first);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    unsigned long long bad[2]; uint32_t first[2];
    (void)hipMemcpy(bad, d_bad, 16, hipMemcpyDeviceToHost); (void)hipMemcpy(first, d_first, 8, hipMemcpyDeviceToHost);
    printf("quotient bit mismatches (numerators >= 2^-63 or 0): %llu   stored-integer mismatches (all numerators): %llu   (%.1f s)", bad[0], bad[1], ms / 1e3);
    if (bad[1]) printf("   e.g. q bits %08x, numerator bits %08x", first[0], first[1]);
    printf("\n");
    return (bad[0] || bad[1]) ? 1 : 0;
}
