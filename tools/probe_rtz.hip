// probe_rtz.hip -- does v_cvt_pk_u8_f32 follow MODE.fp_round?  (and rate of v_max/v_min)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
__global__ void k_cvt(const float *in, unsigned *out, unsigned *out2, int n)
{
    int i = threadIdx.x;
    float v = i < n ? in[i] : 0.f;
    __builtin_amdgcn_s_setreg(2049, 3);   // hwreg(MODE, 0, 2) = FP32 round mode -> toward zero
    unsigned r = __builtin_amdgcn_cvt_pk_u8_f32(v, 0u, 0u);
    __builtin_amdgcn_s_setreg(2049, 0);
    float s = v + 0.3f;                   // must round to nearest again
    if (i < n) { out[i] = r; out2[i] = __float_as_uint(s); }
}
int main()
{
    std::vector<float> v = {0.0f, 0.4f, 0.5f, 0.6f, 0.999f, 1.5f, 2.5f, 3.5f, 3.99999f, 254.5f, 254.999f, 255.0f, 255.4f,
                            255.5f, 255.999f, 256.0f, 300.0f, 1e9f, -0.4f, -0.6f, -0.9999f, -1.0f, -3.0f, 127.5f, 128.5f};
    float *d_in; unsigned *d_out, *d_out2;
    (void)hipMalloc(&d_in, 1024); (void)hipMalloc(&d_out, 1024); (void)hipMalloc(&d_out2, 1024);
    (void)hipMemcpy(d_in, v.data(), v.size() * 4, hipMemcpyHostToDevice);
    k_cvt<<<1, 64>>>(d_in, d_out, d_out2, (int)v.size());
    std::vector<unsigned> o(v.size()), o2(v.size());
    (void)hipMemcpy(o.data(), d_out, v.size() * 4, hipMemcpyDeviceToHost);
    (void)hipMemcpy(o2.data(), d_out2, v.size() * 4, hipMemcpyDeviceToHost);
    int bad = 0;
    for (size_t i = 0; i < v.size(); ++i) {
        float want = v[i] < 0 ? 0 : (v[i] > 255 ? 255 : v[i]);
        unsigned w = (unsigned)want;
        float s = v[i] + 0.3f;
        unsigned sb; memcpy(&sb, &s, 4);
        printf("  %14.5f -> %3u (trunc-clamp %3u) %s | add rn ok: %d\n", v[i], o[i] & 0xff, w, (o[i] & 0xff) == w ? "ok" : "MISMATCH", sb == o2[i]);
        bad += (o[i] & 0xff) != w;
    }
    printf("mismatches: %d\n", bad);
    return 0;
}
