// kerr_rk4_step_fast against kerr_rk4_step_fast_pk on random far-field states: which output component differs, how often
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cmath>
#include <vector>
#include "../../light-path-tracer_amd/csrc/lt_device.hpp"
using namespace lt;
__global__ void k(KerrConsts<float> kc, const float *in, uint32_t *out, float *vals, int n)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    State5<float> y; y.r = in[6 * i]; y.th = in[6 * i + 1]; y.ph = 0.1f; y.pr = in[6 * i + 2]; y.pth = in[6 * i + 3];
    RayConsts<float> rc = make_ray_consts(kc, in[6 * i + 4], false);
    float mr, md, mr2, md2;
    State5<float> a = kerr_rk4_step_fast(kc, rc, y, rc.hb, mr, md);
    float sy, cy; M<float>::sincos(y.th, sy, cy);
    f32x2 cs_next;
    State5<float> b = kerr_rk4_step_fast_pk(kc, rc, y, (f32x2){cy, sy}, rc.hb, mr2, md2, cs_next);
    auto ne = [](float x, float z) { return __float_as_uint(x) != __float_as_uint(z); };
    out[i] = (ne(a.r, b.r) ? 1u : 0u) | (ne(a.th, b.th) ? 2u : 0u) | (ne(a.ph, b.ph) ? 4u : 0u) | (ne(a.pr, b.pr) ? 8u : 0u) | (ne(a.pth, b.pth) ? 16u : 0u) |
             (ne(mr, mr2) ? 32u : 0u) | (ne(md, md2) ? 64u : 0u);
    // one right-hand side alone
    float s, c; M<float>::sincos(y.th, s, c);
    float dr, dth, dph, dpr, dpth;
    kerr_rhs_sc<float, false>(kc, rc, y.r, s, c, y.pr, y.pth, dr, dth, dph, dpr, dpth);
    PkStage g = kerr_rhs_pk(kc, rc, y.r, (f32x2){s, c}, (f32x2){y.pr, y.pth});
    out[n + i] = (ne(dr, g.dr) ? 1u : 0u) | (ne(dth, g.dth) ? 2u : 0u) | (ne(dph, g.dph) ? 4u : 0u) | (ne(dpr, g.dp.x) ? 8u : 0u) | (ne(dpth, g.dp.y) ? 16u : 0u);
    // one rotation alone
    float d = 0.01f * y.pth, s1, c1;
    sincos_shift<float, false>(y.th, s, c, d, s1, c1);
    f32x2 sc1 = sincos_shift_pk((f32x2){c, s}, d);
    out[2 * n + i] = (ne(s1, sc1.x) ? 1u : 0u) | (ne(c1, sc1.y) ? 2u : 0u);
}
int main()
{
    const int n = 1 << 20;
    std::vector<float> in(6 * n);
    uint64_t s = 88172645463325252ull;
    auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return (double)(s >> 11) / 9007199254740992.0; };
    for (int i = 0; i < n; ++i) { in[6 * i] = 8.0 + 80.0 * rnd(); in[6 * i + 1] = 0.2 + 2.7 * rnd(); in[6 * i + 2] = -1.0 + 2.0 * rnd(); in[6 * i + 3] = -8.0 + 16.0 * rnd(); in[6 * i + 4] = -10.0 + 20.0 * rnd(); }
    KerrConsts<float> kc{};
    kc.M = 1.0f; kc.a = 0.9f; kc.a2 = 0.81f; kc.two_M = 2.0f; kc.r_cut = 1.437f; kc.r_capture = 1.45f; kc.r_escape = 100.0f; kc.r_obs = 50.0f; kc.theta_obs = 1.5707964f;
    kc.lambda_max = 5000.0f; kc.h_max = 1.0f; kc.rc4 = 5.8f; kc.rc2 = 2.9f; kc.rc12 = 1.74f;
    float *din; uint32_t *dout; float *dv;
    (void)hipMalloc(&din, in.size() * 4); (void)hipMalloc(&dout, 3 * n * 4); (void)hipMalloc(&dv, 4);
    (void)hipMemcpy(din, in.data(), in.size() * 4, hipMemcpyHostToDevice);
    k<<<n / 256, 256>>>(kc, din, dout, dv, n);
    std::vector<uint32_t> out(3 * n);
    (void)hipMemcpy(out.data(), dout, 3 * n * 4, hipMemcpyDeviceToHost);
    const char *names[3] = {"step (r th ph pr pth min_r max_d)", "rhs (dr dth dph dpr dpth)", "rotation (s c)"};
    for (int p = 0; p < 3; ++p) {
        long cnt[7] = {0, 0, 0, 0, 0, 0, 0}, any = 0;
        for (int i = 0; i < n; ++i) { uint32_t f = out[p * n + i]; if (f) ++any; for (int j = 0; j < 7; ++j) if (f & (1u << j)) ++cnt[j]; }
        printf("%s: %ld of %d differ; per component %ld %ld %ld %ld %ld %ld %ld\n", names[p], any, n, cnt[0], cnt[1], cnt[2], cnt[3], cnt[4], cnt[5], cnt[6]);
    }
    return 0;
}
