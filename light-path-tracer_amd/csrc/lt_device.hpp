// lt_device.hpp -- __device__ building blocks of the gfx950 ray integrators.
//
// Written for CDNA4 wave64: everything a ray needs lives in registers, the metric terms are
// inlined, there is no LDS and no cross-lane traffic in the hot loop.
//
// What each block computes is fixed by the reference (cited per function); HOW it is computed is
// not a transcription.  The integrate kernel is VALU-issue bound, and on gfx950 (measured with
// tools/issue_probe.py) every VALU instruction costs ~2.1 issue cycles per wave
// when FMA-class and "second pipe" instructions (min/max/cmp/cvt/select, anything with an SGPR
// operand) are mixed, a transcendental (v_rcp_f32 ...) costs ~13, and a packed v_pk_* costs two
// plain ones.  So the design rule is: fewest instructions, fewest reciprocals.  The Kerr
// right-hand side is re-derived from the separable form of the Hamiltonian (one merged reciprocal,
// ~75 instructions instead of the reference's 186 flops + 14 divides) while remaining the exact
// gradient of the same H, so it agrees with the reference for any state, on- or off-shell
// (tests/test_gpu_parity.py::test_kerr_rhs_probe_matches_reference).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace lt {

// true if the predicate holds in any active lane: the wave-uniform test behind every fast path.  The builtin
// reads the compare's lane mask directly (HIP's __ballot(int) first materialises the bool in a VGPR and
// compares it again: two extra VALU instructions per test).
__device__ __forceinline__ bool wave_any(bool p) { return __builtin_amdgcn_ballot_w64(p) != 0ull; }
// true if the predicate holds in every ACTIVE lane.  (Written on the predicate itself, not as !wave_any(!p): when p is
// also used for per-lane selects the compiler keeps its lane mask, and negating it first costs a 0/1 materialisation
// and a compare -- two VALU instructions per test.)
__device__ __forceinline__ bool wave_all(bool p) { return __builtin_amdgcn_ballot_w64(p) == __builtin_amdgcn_ballot_w64(true); }


// ---------------------------------------------------------------------------------------
// scalar math wrappers
// ---------------------------------------------------------------------------------------
template <typename T> struct M;

template <> struct M<float> {
    // 1/x: hardware reciprocal (1 ulp) + one Newton step
    static __device__ __forceinline__ float rcp(float x)
    {
        float y = __builtin_amdgcn_rcpf(x);
        float e = __builtin_fmaf(-x, y, 1.0f);
        return __builtin_fmaf(y, e, y);
    }
    // 1/x for positive normal x WITHOUT the transcendental unit.  Measured on gfx950 (lt_piece_probe):
    // one v_rcp_f32 inside this kernel's FMA-dense stream costs ~70 SIMD cycles, whatever the
    // occupancy and whether or not anything depends on it, against ~13 for six FMA-class instructions.
    // Bit trick seed (relative error <= 5.1 %), one cubic and one quadratic Newton step: error 1.7e-8
    // before rounding, i.e. correctly rounded to within 1 ulp like rcp + Newton.
    static __device__ __forceinline__ float rcp_pos(float x)
    {
        float y = __uint_as_float(0x7EF311C0u - __float_as_uint(x));
        float e = __builtin_fmaf(-x, y, 1.0f);
        y = __builtin_fmaf(y, __builtin_fmaf(e, e, e), y);
        e = __builtin_fmaf(-x, y, 1.0f);
        return __builtin_fmaf(y, e, y);
    }
    static __device__ __forceinline__ float fma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
    static __device__ __forceinline__ float abs(float x) { return __builtin_fabsf(x); }
    static __device__ __forceinline__ float max(float a, float b) { return __builtin_fmaxf(a, b); }
    static __device__ __forceinline__ float min(float a, float b) { return __builtin_fminf(a, b); }
    // sin^2 floored at 1e-15 (metrics.py:236-237): below float resolution unless sin^2 < 1e-8, so an
    // add does the job of the max without leaving the FMA pipe.
    static __device__ __forceinline__ float sin2_floor(float s) { return __builtin_fmaf(s, s, 1e-15f); }
    // sin and cos of an angle of modest size (|x| < ~1e4; the polar angle of a ray stays within a
    // few pi).  Cody-Waite reduction by pi/2 in three pieces, single-precision minimax polynomials
    // on [-pi/4, pi/4], quadrant fix-up with integer ops: ~25 VALU ops for both values, ~1 ulp.
    static __device__ __forceinline__ void sincos(float x, float &s, float &c)
    {
        const float TWO_OVER_PI = 0.636619772367581343f;
        const float P1 = 1.5703125f; // pi/2 split: 8 + 11 + 24 significant bits
        const float P2 = 4.837512969970703125e-4f;
        const float P3 = 7.54978995489188e-8f;
        float kf = __builtin_rintf(x * TWO_OVER_PI);
        float y = __builtin_fmaf(-kf, P1, x);
        y = __builtin_fmaf(-kf, P2, y);
        y = __builtin_fmaf(-kf, P3, y);
        int k = (int)kf;
        float z = y * y;
        float ps = __builtin_fmaf(z, -1.9515295891e-4f, 8.3321608736e-3f);
        ps = __builtin_fmaf(ps, z, -1.6666654611e-1f);
        float sy = __builtin_fmaf(ps * z, y, y);
        float pc = __builtin_fmaf(z, 2.443315711809948e-5f, -1.388731625493765e-3f);
        pc = __builtin_fmaf(pc, z, 4.166664568298827e-2f);
        float cy = __builtin_fmaf(pc * z, z, __builtin_fmaf(-0.5f, z, 1.0f));
        bool swap = k & 1;
        float ss = swap ? cy : sy;
        float cc = swap ? sy : cy;
        // sign: sin flips for k = 2,3 (mod 4), cos for k = 1,2
        uint32_t sbit = ((uint32_t)k << 30) & 0x80000000u;
        uint32_t cbit = ((uint32_t)(k + 1) << 30) & 0x80000000u;
        s = __uint_as_float(__float_as_uint(ss) ^ sbit);
        c = __uint_as_float(__float_as_uint(cc) ^ cbit);
    }
    // sin and cos of a SMALL angle (|d| <= 0.25): Taylor, 7 FMA-class instructions, error < 1.3e-8
    static __device__ __forceinline__ void sincos_small(float d, float &s, float &c)
    {
        float z = d * d;
        float ps = __builtin_fmaf(z, 8.3333333333e-3f, -1.6666666667e-1f);
        s = __builtin_fmaf(ps * z, d, d);
        float pc = __builtin_fmaf(z, -1.3888888889e-3f, 4.1666666667e-2f);
        pc = __builtin_fmaf(pc, z, -0.5f);
        c = __builtin_fmaf(pc, z, 1.0f);
    }
    // true unless x is NaN or +-inf
    static __device__ __forceinline__ bool finite(float x) { return __builtin_fabsf(x) < __builtin_inff(); }
};

// sin and cos of a float64 angle of modest size without libm's generic range reduction: Cody-Waite
// by pi/2 (fdlibm's pio2_1 + pio2_1t split, a double-double remainder y0 + y1 good to |x| ~ 1e5) and
// the fdlibm kernel polynomials on [-pi/4, pi/4] (< 1 ulp).  ~40 float64 FMA-class instructions, no
// branches, no scratch -- OCML's sincos carries a Payne-Hanek path and costs registers and time.
__device__ __forceinline__ void sincos_f64(double x, double &s, double &c)
{
    const double TWO_OVER_PI = 6.36619772367581382433e-01;
    const double P1 = 1.57079632673412561417e+00, P1T = 6.07710050650619224932e-11;
    double kf = __builtin_rint(x * TWO_OVER_PI);
    double r = __builtin_fma(-kf, P1, x);
    double w = kf * P1T;
    double y0 = r - w;
    double y1 = (r - y0) - w;
    int k = (int)kf;
    double z = y0 * y0;
    const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03, S3 = -1.98412698298579493134e-04,
                 S4 = 2.75573137070700676789e-06, S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
    double ps = __builtin_fma(z, S6, S5);
    ps = __builtin_fma(z, ps, S4);
    ps = __builtin_fma(z, ps, S3);
    ps = __builtin_fma(z, ps, S2);
    ps = __builtin_fma(z, ps, S1);
    double sy = __builtin_fma(z * y0, ps, y0);
    const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03, C3 = 2.48015872894767294178e-05,
                 C4 = -2.75573143513906633035e-07, C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
    double pc = __builtin_fma(z, C6, C5);
    pc = __builtin_fma(z, pc, C4);
    pc = __builtin_fma(z, pc, C3);
    pc = __builtin_fma(z, pc, C2);
    pc = __builtin_fma(z, pc, C1);
    double cy = __builtin_fma(z * z, pc, __builtin_fma(-0.5, z, 1.0));
    double sr = __builtin_fma(y1, cy, sy), cr = __builtin_fma(-y1, sy, cy); // first-order in the low word
    bool swap = k & 1;
    double ss = swap ? cr : sr, cc = swap ? sr : cr;
    s = (k & 2) ? -ss : ss;
    c = ((k + 1) & 2) ? -cc : cc;
}

template <> struct M<double> {
    static __device__ __forceinline__ double rcp(double x) { return 1.0 / x; }
    // 1/x for positive normal x: the hardware seed and two Newton steps -- the core of the compiler's IEEE division
    // (~12 instructions) without its scaling, residual correction and special-case fix-up; within an ulp
    static __device__ __forceinline__ double rcp_pos(double x)
    {
        double y = __builtin_amdgcn_rcp(x);
        double e = __builtin_fma(-x, y, 1.0);
        y = __builtin_fma(y, e, y);
        e = __builtin_fma(-x, y, 1.0);
        return __builtin_fma(y, e, y);
    }
    static __device__ __forceinline__ double fma(double a, double b, double c) { return __builtin_fma(a, b, c); }
    static __device__ __forceinline__ double abs(double x) { return __builtin_fabs(x); }
    static __device__ __forceinline__ double max(double a, double b) { return __builtin_fmax(a, b); }
    static __device__ __forceinline__ double min(double a, double b) { return __builtin_fmin(a, b); }
    static __device__ __forceinline__ double sin2_floor(double s) { return __builtin_fmax(s * s, 1e-15); }
    static __device__ __forceinline__ void sincos(double x, double &s, double &c) { sincos_f64(x, s, c); }
    // |d| <= 0.25: Taylor to d^11 / d^12, error < 3e-18
    static __device__ __forceinline__ void sincos_small(double d, double &s, double &c)
    {
        double z = d * d;
        double ps = __builtin_fma(z, -2.50521083854417188e-08, 2.75573192239858907e-06);
        ps = __builtin_fma(ps, z, -1.98412698412698413e-04);
        ps = __builtin_fma(ps, z, 8.33333333333333333e-03);
        ps = __builtin_fma(ps, z, -1.66666666666666667e-01);
        s = __builtin_fma(ps * z, d, d);
        double pc = __builtin_fma(z, 2.08767569878680990e-09, -2.75573192239858907e-07);
        pc = __builtin_fma(pc, z, 2.48015873015873016e-05);
        pc = __builtin_fma(pc, z, -1.38888888888888889e-03);
        pc = __builtin_fma(pc, z, 4.16666666666666667e-02);
        pc = __builtin_fma(pc, z, -0.5);
        c = __builtin_fma(pc, z, 1.0);
    }
    static __device__ __forceinline__ bool finite(double x) { return __builtin_fabs(x) < __builtin_inf(); }
};

// ---------------------------------------------------------------------------------------
// Kerr
// ---------------------------------------------------------------------------------------
// Wave-uniform constants of one render (kernel argument -> SGPRs).
template <typename T> struct KerrConsts {
    T M, a, a2;
    T two_M;
    T r_cut;       // 1.001 r_plus: RHS returns zero at or inside (metrics.py:228-231)
    T r_capture;   // 1.01 r_plus  (metrics.py:579)
    T r_escape;    // 2 r_obs      (metrics.py:580)
    T r_obs, theta_obs;
    T lambda_max;
    T h_max;       // 1.0 (metrics.py:677)
    T rc4, rc2, rc12; // r_capture * 4, * 2, * 1.2 (metrics.py:606-611)
};

// A VALU instruction with an SGPR source runs on the half-rate pipe on gfx950 (v_fma_f32 with an
// SGPR operand: 4.1 issue cycles against 2.2; tools/issue_probe.py).  The constants the right-hand
// side multiplies with are therefore copied into VGPRs once per kernel; the empty asm makes the
// compiler forget that they are wave-uniform.
#ifndef LT_PIN_CONSTS
#define LT_PIN_CONSTS 1
#endif
template <typename T> __device__ __forceinline__ void pin_consts(KerrConsts<T> &k)
{
#if LT_PIN_CONSTS
    asm volatile("" : "+v"(k.a), "+v"(k.a2), "+v"(k.two_M));
#endif
}

// Per-ray constants: the conserved p_phi = L (p_t = -1 throughout, metrics.py:187), what follows
// from it, and the ray's step-size table (axis-refine rays use tighter caps, metrics.py:591-611).
template <typename T> struct RayConsts {
    T L;
    T c_P;   // a^2 - a L   ->  P = r^2 + c_P        (P = (r^2 + a^2) E - a L, E = 1)
    T c_W;   // -2 a L      ->  W = L^2/s2 + c_W + a^2 s2
    T hb;        // base step: h_max, or min(h_max, 0.5) on axis-refine rays (metrics.py:591-593)
    bool refine; // axis-refine ray: tighter radius-band caps and retry floor
};

template <typename T>
__device__ __forceinline__ RayConsts<T> make_ray_consts(const KerrConsts<T> &k, T L, bool refine)
{
    RayConsts<T> rc;
    rc.L = L;
    rc.c_P = M<T>::fma(-k.a, L, k.a2);
    rc.c_W = T(-2) * k.a * L;
    rc.hb = refine ? M<T>::min(k.h_max, T(0.5)) : k.h_max;
    rc.refine = refine;
    return rc;
}

// Hamilton's equations for H = 1/2 g^{mu nu} p_mu p_nu in Boyer-Lindquist coordinates, reduced
// 5-D state (r, theta, phi, p_r, p_theta); the function the reference spends >95 % of its time in
// (metrics.py:221-303).  With E = -p_t = 1:
//     2 Sigma H = F = Delta p_r^2 + p_theta^2 + W(theta) - P(r)^2 / Delta,
//     W = L^2/sin^2 - 2 a L + a^2 sin^2,   P = r^2 + a^2 - a L,
// so dx/dlambda = dH/dp and dp/dlambda = -dH/dx = -(F_x - 2H Sigma_x) / (2 Sigma).  The 2H term is
// zero on a null geodesic but is kept: the reference differentiates the full H, and integration
// error drives H slightly off zero.  The three reciprocals 1/Sigma, 1/Delta, 1/sin^2 come from ONE
// hardware reciprocal of their product.  At or inside r_cut the reference returns zeros
// (metrics.py:228-231): every output carries the factor 1/Sigma, so masking that one factor (and
// evaluating at max(r, r_cut) so nothing overflows) does it.
//
// CHECKED = false leaves the r <= r_cut handling out: for callers that know, or verify afterwards, that
// the radius is outside (kerr_rk4_step_fast below).  Arithmetic otherwise identical.
//
// The three velocities are products -- dr = Delta (p_r / Sigma), dtheta = p_theta / Sigma, dphi = (1 / Sigma) u -- and the
// last stage of an RK4 step adds them to a running sum: the compiler (-ffp-contract=fast) fuses that product and sum,
// for the stage whose result has no other use.  kerr_rhs_parts hands out the factors so that the step can write that
// fma itself (kerr_rk4_step_impl): the rounding then no longer depends on what a compiler decides to fuse, and the
// packed form of the step (kerr_rk4_step_fast_pk) can be made to round identically.
template <typename T> struct KerrRhsParts {
    T Delta, iSpr; // dr = Delta * iSpr
    T iS;          // dtheta = p_theta * iS
    T u;           // dphi = iS * u
    T dpr, dpth;
};

template <typename T, bool CHECKED = true>
__device__ __forceinline__ KerrRhsParts<T> kerr_rhs_parts(const KerrConsts<T> &k, const RayConsts<T> &rc, T r_in, T s, T c, T pr, T pth)
{
    bool inside = CHECKED && r_in <= k.r_cut;
    const bool any_inside = CHECKED && wave_any(inside); // almost never: only a stage of the last steps before capture
    T r = r_in;
    if (__builtin_expect(any_inside, 0)) r = inside ? k.r_cut : r_in;
    T s2 = M<T>::sin2_floor(s);
    // (explicit fma, not r * r + a2 / L * is2 - a below: under -ffp-contract=fast the compiler decides per template
    // instantiation whether to fuse such an expression, and the checked and the branch-free variant of this function
    // must round identically -- found in round 2 when a variant that let unflagged lanes keep the checked step's
    // result differed from the other schedule in 41 of 4.2 M pixels by one ulp)
    T ra = M<T>::fma(r, r, k.a2);
    T Sigma = M<T>::fma(-k.a2, s2, ra);          // r^2 + a^2 cos^2 as r^2 + a^2 - a^2 sin^2 (the 1e-15 floor is far below an ulp)
    T Delta = M<T>::fma(-k.two_M, r, ra);
    T SD = Sigma * Delta;
    T t = M<T>::rcp_pos(SD * s2); // > 0: r >= r_cut > r_plus so Delta > 0, Sigma > 0, s2 >= 1e-15
    T iS = (Delta * s2) * t;
    T iD = (Sigma * s2) * t;
    T is2 = SD * t;
    if (__builtin_expect(any_inside, 0)) iS = inside ? T(0) : iS;
    T P = M<T>::fma(r, r, rc.c_P);
    T q = P * iD;
    T Lis2 = rc.L * is2;
    T pr2 = pr * pr;
    KerrRhsParts<T> o;
    o.Delta = Delta;
    o.iSpr = iS * pr;
    o.iS = iS;
    o.u = M<T>::fma(k.a, q, M<T>::fma(rc.L, is2, -k.a));
    // (dropping the 2H terms -- zero on a null geodesic -- would save instructions, but the reference's RK4
    // solution drifts off-shell at h = 1 and they matter: measured median |d final_alpha| 5.6e-6 without them
    // against 4.4e-7 with them, and p99 1e-1 against 2e-5)
    T W = M<T>::fma(rc.L, Lis2, M<T>::fma(k.a2, s2, rc.c_W));
    T F = M<T>::fma(Delta, pr2, M<T>::fma(pth, pth, M<T>::fma(-P, q, W)));
    T H2 = F * iS;
    // dp_r = -(F_r - 2H Sigma_r) / (2 Sigma) with F_r = (2r - 2M)(q^2 + p_r^2) - 4 r q, Sigma_r = 2r; the factor
    // 1/2 goes into the bracket:  -(1/Sigma) [ (r - M)(q^2 + p_r^2) - r (2q + 2H) ]
    T rt = r * M<T>::fma(T(2), q, H2);
    o.dpr = -iS * M<T>::fma(r - k.M, M<T>::fma(q, q, pr2), -rt);
    // dp_theta likewise: F_theta = 2 s c (a^2 - L^2/s^4), Sigma_theta = -2 a^2 s c:  -(1/Sigma) s c [a^2 (1 + 2H) - (L/s^2)^2]
    o.dpth = -iS * ((s * c) * M<T>::fma(H2, k.a2, M<T>::fma(-Lis2, Lis2, k.a2)));
    return o;
}

template <typename T, bool CHECKED = true>
__device__ __forceinline__ void kerr_rhs_sc(const KerrConsts<T> &k, const RayConsts<T> &rc, T r_in, T s, T c, T pr, T pth,
                                            T &dr, T &dth, T &dph, T &dpr, T &dpth)
{
    const KerrRhsParts<T> o = kerr_rhs_parts<T, CHECKED>(k, rc, r_in, s, c, pr, pth);
    dr = o.Delta * o.iSpr;
    dth = pth * o.iS;
    dph = o.iS * o.u;
    dpr = o.dpr;
    dpth = o.dpth;
}

template <typename T>
__device__ __forceinline__ void kerr_rhs(const KerrConsts<T> &k, const RayConsts<T> &rc, T r, T th, T pr, T pth,
                                         T &dr, T &dth, T &dph, T &dpr, T &dpth)
{
    T s, c;
    M<T>::sincos(th, s, c);
    kerr_rhs_sc(k, rc, r, s, c, pr, pth, dr, dth, dph, dpr, dpth);
}

// x^(-1/5) for the float64 step controllers (dense tracks: err^(-1/5), where solve_ivp's own libm differs from ours in
// the last bits too; DP45 with the reference's controller: err_norm^(-0.2)).  A float32 seed from the hardware log2 / exp2 (relative error ~1e-6) and two Newton steps on
// y^-5 = x,  y <- y + y (1 - x y^5) / 5  (quadratic: 1e-6 -> 3e-12 -> 1e-22, then the rounding of the evaluation,
// ~1 ulp): 18 instructions where exp2(-0.2 log2(x)) in float64 library code takes ~110.  Outside [1e-6, 1e4] the
// controller clamps the factor anyway (0.9 y > 10, or < 0.2), so any value on the right side of the clamp does; a NaN
// comes out as a NaN, as before.
__device__ __forceinline__ double pow_m02(double x)
{
    const float xf = (float)x;
    double y = (double)__builtin_amdgcn_exp2f(-0.2f * __builtin_amdgcn_logf(xf));
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const double y2 = y * y, y5 = (y2 * y2) * y;
        y = __builtin_fma(y, 0.2 * __builtin_fma(-x, y5, 1.0), y);
    }
    y = x < 1e-6 ? 100.0 : y;
    y = x > 1e4 ? 0.1 : y;
    return y;
}

// sin and cos of th0 + d from (s0, c0) = sincos(th0): the stage states of a Runge-Kutta step differ from
// the step's base state by h * (a small angular velocity), so the 25-instruction argument reduction +
// polynomials shrink to an 11-instruction rotation.  The wave falls back to the full evaluation in the
// rare step where some lane's |d| exceeds 0.25 rad (wave-uniform branch).
template <typename T, bool CHECKED = true>
__device__ __forceinline__ void sincos_shift(T th0, T s0, T c0, T d, T &s, T &c)
{
    T sd, cd;
    M<T>::sincos_small(d, sd, cd);
    s = M<T>::fma(s0, cd, c0 * sd);
    c = M<T>::fma(c0, cd, -(s0 * sd));
    bool big = CHECKED && M<T>::abs(d) > T(0.25);
    if (__builtin_expect(CHECKED && wave_any(big), 0)) {
        // which formula a lane uses depends on ITS angle only, never on its neighbours: results stay
        // bit-identical however rays are grouped into waves (direct vs queue schedule, partitions)
        T sf, cf;
        M<T>::sincos(th0 + d, sf, cf);
        s = big ? sf : s;
        c = big ? cf : c;
    }
}

template <typename T> struct State5 {
    T r, th, ph, pr, pth;
};

// Classic RK4 step, metrics.py:306-323 (phi does not enter the right-hand side, so the stage states
// carry only r, theta, p_r, p_theta).
//
// CHECKED = true: every stage handles r <= r_cut and a stage angle offset beyond 0.25 rad on the spot (one
// wave-uniform branch each: seven per step).  CHECKED = false: no such branches; the step reports the
// smallest stage radius and the largest stage offset instead, so that the caller can tell afterwards
// whether either case occurred (then the unchecked result is wrong for that lane and must be replaced by
// the checked one) -- one test per step instead of seven, for the same results.
template <typename T, bool CHECKED>
__device__ __forceinline__ State5<T> kerr_rk4_step_impl(const KerrConsts<T> &k, const RayConsts<T> &rc,
                                                        const State5<T> &y, T h, T &min_r, T &max_d)
{
    T k_r, k_th, k_ph, k_pr, k_pth;
    T a_r, a_th, a_ph, a_pr, a_pth; // running k1 + 2 k2 + 2 k3 + k4
    T s0, c0, s, c;
    M<T>::sincos(y.th, s0, c0); // the only full sincos of the step; stages rotate it (sincos_shift)
    kerr_rhs_sc<T, CHECKED>(k, rc, y.r, s0, c0, y.pr, y.pth, k_r, k_th, k_ph, k_pr, k_pth);
    a_r = k_r; a_th = k_th; a_ph = k_ph; a_pr = k_pr; a_pth = k_pth;
    T hh = T(0.5) * h;
    T t_r = M<T>::fma(hh, k_r, y.r), t_pr = M<T>::fma(hh, k_pr, y.pr), t_pth = M<T>::fma(hh, k_pth, y.pth);
    T d2 = hh * k_th;
    T r2 = t_r;
    sincos_shift<T, CHECKED>(y.th, s0, c0, d2, s, c);
    kerr_rhs_sc<T, CHECKED>(k, rc, t_r, s, c, t_pr, t_pth, k_r, k_th, k_ph, k_pr, k_pth);
    a_r = M<T>::fma(T(2), k_r, a_r); a_th = M<T>::fma(T(2), k_th, a_th); a_ph = M<T>::fma(T(2), k_ph, a_ph);
    a_pr = M<T>::fma(T(2), k_pr, a_pr); a_pth = M<T>::fma(T(2), k_pth, a_pth);
    t_r = M<T>::fma(hh, k_r, y.r); t_pr = M<T>::fma(hh, k_pr, y.pr); t_pth = M<T>::fma(hh, k_pth, y.pth);
    T d3 = hh * k_th;
    T r3 = t_r;
    sincos_shift<T, CHECKED>(y.th, s0, c0, d3, s, c);
    kerr_rhs_sc<T, CHECKED>(k, rc, t_r, s, c, t_pr, t_pth, k_r, k_th, k_ph, k_pr, k_pth);
    a_r = M<T>::fma(T(2), k_r, a_r); a_th = M<T>::fma(T(2), k_th, a_th); a_ph = M<T>::fma(T(2), k_ph, a_ph);
    a_pr = M<T>::fma(T(2), k_pr, a_pr); a_pth = M<T>::fma(T(2), k_pth, a_pth);
    t_r = M<T>::fma(h, k_r, y.r); t_pr = M<T>::fma(h, k_pr, y.pr); t_pth = M<T>::fma(h, k_pth, y.pth);
    T d4 = h * k_th;
    sincos_shift<T, CHECKED>(y.th, s0, c0, d4, s, c);
    const KerrRhsParts<T> k4 = kerr_rhs_parts<T, CHECKED>(k, rc, t_r, s, c, t_pr, t_pth);
    if (!CHECKED) {
        // (a NaN stage value is ignored by min / max; it makes the step's result non-finite in either variant)
        min_r = M<T>::min(M<T>::min(r2, r3), t_r);
        max_d = M<T>::max(M<T>::max(M<T>::abs(d2), M<T>::abs(d3)), M<T>::abs(d4));
    }
    T h6 = h * T(1.0 / 6.0);
    State5<T> o;
    // (k1 + 2 k2 + 2 k3) + k4 with the velocity products of the last stage fused into the sum, see kerr_rhs_parts
    o.r = M<T>::fma(h6, M<T>::fma(k4.Delta, k4.iSpr, a_r), y.r);
    o.th = M<T>::fma(h6, M<T>::fma(t_pth, k4.iS, a_th), y.th);
    o.ph = M<T>::fma(h6, M<T>::fma(k4.iS, k4.u, a_ph), y.ph);
    o.pr = M<T>::fma(h6, a_pr + k4.dpr, y.pr);
    o.pth = M<T>::fma(h6, a_pth + k4.dpth, y.pth);
    return o;
}

template <typename T>
__device__ __forceinline__ State5<T> kerr_rk4_step(const KerrConsts<T> &k, const RayConsts<T> &rc,
                                                   const State5<T> &y, T h)
{
    T unused_r, unused_d;
    return kerr_rk4_step_impl<T, true>(k, rc, y, h, unused_r, unused_d);
}

// The branch-free variant; valid for a lane iff min_r > k.r_cut and max_d <= 0.25 (kerr_rk4_step_ok).
template <typename T>
__device__ __forceinline__ State5<T> kerr_rk4_step_fast(const KerrConsts<T> &k, const RayConsts<T> &rc,
                                                        const State5<T> &y, T h, T &min_r, T &max_d)
{
    return kerr_rk4_step_impl<T, false>(k, rc, y, h, min_r, max_d);
}

// ---------------------------------------------------------------------------------------
// The same step for a wavefront that is ALONE on its SIMD (the last waves of a launch, hosting the few very long
// rays; k_kerr_direct's ghost-lane phase).  Such a wave issues one instruction per ~4.8 cycles whatever the
// instruction is (tools/issue_probe.py), and a packed v_pk_fma_f32 / v_pk_mul_f32 -- two float32 operations on a
// 64-bit register pair, each half of each source chosen by op_sel -- costs it 5.3: wherever two operations of the
// step take their operands from the same pairs they share one issue slot.  (With two or more waves per SIMD a
// packed instruction costs two issue slots and this form gains nothing: the bulk keeps the scalar step.)
// Every operation below is one of the scalar step's, with the same operands: a product x * y is written
// fma(x, y, -0) where its slot-mate is an fma -- identical bits, including the sign of a zero product -- so the
// result is bit-identical to kerr_rk4_step_fast (tests/test_gpu_ghost_lanes.py runs whole frames through either).
// Pairs: [ps, pc] polynomials of the stage rotation, [s, c], [s^2, s c], [r^2 + a^2, P], [Sigma, Delta] s^2 -> [1/Delta, 1/Sigma]
// after the one reciprocal, [L/s^2, .], [p_r/Sigma, dtheta], [dphi, 2H], [dp_r, dp_theta], and the RK4 bookkeeping of
// the (r, theta) and (p_r, p_theta) components.
typedef float f32x2 __attribute__((ext_vector_type(2)));

struct PkStage {
    float dr, dth, dph; // LAST: unset
    f32x2 dp;           // [dp_r, dp_theta]
    float Delta, iSpr, iS, u; // LAST only: the factors of dr, dtheta, dphi (KerrRhsParts)
};

__device__ __forceinline__ f32x2 pk_bc(float x) { return (f32x2){x, x}; }
__device__ __forceinline__ f32x2 pk_fma(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }

// right-hand side at (r, [sin, cos], [p_r, p_theta]); kerr_rhs_sc<float, false>, LAST: kerr_rhs_parts<float, false>
template <bool LAST = false>
__device__ __forceinline__ PkStage kerr_rhs_pk(const KerrConsts<float> &k, const RayConsts<float> &rc, float r, f32x2 sc, f32x2 p)
{
#pragma clang fp contract(off)
    PkStage o;
    const f32x2 s2sc = pk_fma(pk_bc(sc.x), sc, (f32x2){1e-15f, -0.0f});          // [s^2 + 1e-15, s c]
    const float s2 = s2sc.x;
    const f32x2 raP = pk_fma(pk_bc(r), pk_bc(r), (f32x2){k.a2, rc.c_P});        // [r^2 + a^2, P]
    const float ra = raP.x, P = raP.y;
    const float Sigma = __builtin_fmaf(-k.a2, s2, ra);
    const float Delta = __builtin_fmaf(-k.two_M, r, ra);
    const f32x2 SDs2 = (f32x2){Sigma, Delta} * pk_bc(s2);                       // [Sigma s^2, Delta s^2]
    const float SD = Sigma * Delta;
    const float t = M<float>::rcp_pos(SD * s2);
    const f32x2 inv = SDs2 * pk_bc(t);                                          // [1/Delta, 1/Sigma]
    const float iD = inv.x, iS = inv.y;
    const float is2 = SD * t;
    const f32x2 Lu = pk_fma(pk_bc(rc.L), pk_bc(is2), (f32x2){-0.0f, -k.a});     // [L / s^2, L / s^2 - a]
    const float Lis2 = Lu.x;
    const float q = P * iD;
    const float pr2 = p.x * p.x;
    const float u2 = __builtin_fmaf(k.a, q, Lu.y);
    const float W = __builtin_fmaf(rc.L, Lis2, __builtin_fmaf(k.a2, s2, rc.c_W));
    const float F = __builtin_fmaf(Delta, pr2, __builtin_fmaf(p.y, p.y, __builtin_fmaf(-P, q, W)));
    float H2;
    if (LAST) {
        o.Delta = Delta; o.iSpr = iS * p.x; o.iS = iS; o.u = u2;
        H2 = F * iS;
    } else {
        const f32x2 ip = pk_bc(iS) * p;                                         // [p_r / Sigma, p_theta / Sigma]
        o.dr = Delta * ip.x;
        o.dth = ip.y;
        const f32x2 dphH = pk_bc(iS) * (f32x2){u2, F};                          // [dphi, 2H]
        o.dph = dphH.x;
        H2 = dphH.y;
    }
    const float rt = r * __builtin_fmaf(2.0f, q, H2);
    const float b = __builtin_fmaf(r - k.M, __builtin_fmaf(q, q, pr2), -rt);
    const float m2 = __builtin_fmaf(H2, k.a2, __builtin_fmaf(-Lis2, Lis2, k.a2));
    const float scm = s2sc.y * m2;
    o.dp = pk_bc(-iS) * (f32x2){b, scm};                                        // [dp_r, dp_theta]
    return o;
}

// [sin, cos] of th0 + d from cs0 = [cos, sin](th0); sincos_shift<float, false>
__device__ __forceinline__ f32x2 sincos_shift_pk(f32x2 cs0, float d)
{
#pragma clang fp contract(off)
    const float z = d * d;
    f32x2 pp = pk_fma(pk_bc(z), (f32x2){8.3333333333e-3f, -1.3888888889e-3f}, (f32x2){-1.6666666667e-1f, 4.1666666667e-2f});
    pp = pk_fma(pp, pk_bc(z), (f32x2){-0.0f, -0.5f});                            // [ps z, pc]
    const float sd = __builtin_fmaf(pp.x, d, d);
    const float cd = __builtin_fmaf(pp.y, z, 1.0f);
    const f32x2 x = pk_bc(sd) * cs0;                                            // [c0 sd, s0 sd]
    return pk_fma(__builtin_shufflevector(cs0, cs0, 1, 0), pk_bc(cd), (f32x2){x.x, -x.y}); // [s0 cd + c0 sd, c0 cd - s0 sd]
}

// cs0 = [cos, sin](y.th) comes in and [cos, sin] of the new polar angle goes out: the argument reduction of the NEXT
// step's base angle depends on nothing but o.th, which is ready well before the last stage's momenta are, so written
// here the scheduler can run it beside the tail of the last right-hand side instead of at the head of the next step,
// where everything waits for it (a lone wave is bound by the depth of its dependence chains).
__device__ __forceinline__ State5<float> kerr_rk4_step_fast_pk(const KerrConsts<float> &k, const RayConsts<float> &rc,
                                                               const State5<float> &y, f32x2 cs0, float h, float &min_r,
                                                               float &max_d, f32x2 &cs_out)
{
#pragma clang fp contract(off)
    const float s0 = cs0.y, c0 = cs0.x;
    const f32x2 yp = {y.pr, y.pth}, yrt = {y.r, y.th};
    PkStage g = kerr_rhs_pk(k, rc, y.r, (f32x2){s0, c0}, yp);
    f32x2 a_rt = {g.dr, g.dth}, a_p = g.dp; // running k1 + 2 k2 + 2 k3 + k4
    float a_ph = g.dph;
    const float hh = 0.5f * h;
    float t_r = __builtin_fmaf(hh, g.dr, y.r);
    f32x2 t_p = pk_fma(pk_bc(hh), g.dp, yp);
    const float d2 = hh * g.dth;
    const float r2 = t_r;
    g = kerr_rhs_pk(k, rc, t_r, sincos_shift_pk(cs0, d2), t_p);
    a_rt = pk_fma(pk_bc(2.0f), (f32x2){g.dr, g.dth}, a_rt);
    a_ph = __builtin_fmaf(2.0f, g.dph, a_ph);
    a_p = pk_fma(pk_bc(2.0f), g.dp, a_p);
    t_r = __builtin_fmaf(hh, g.dr, y.r);
    t_p = pk_fma(pk_bc(hh), g.dp, yp);
    const float d3 = hh * g.dth;
    const float r3 = t_r;
    g = kerr_rhs_pk(k, rc, t_r, sincos_shift_pk(cs0, d3), t_p);
    a_rt = pk_fma(pk_bc(2.0f), (f32x2){g.dr, g.dth}, a_rt);
    a_ph = __builtin_fmaf(2.0f, g.dph, a_ph);
    a_p = pk_fma(pk_bc(2.0f), g.dp, a_p);
    t_r = __builtin_fmaf(h, g.dr, y.r);
    t_p = pk_fma(pk_bc(h), g.dp, yp);
    const float d4 = h * g.dth;
    g = kerr_rhs_pk<true>(k, rc, t_r, sincos_shift_pk(cs0, d4), t_p);
    min_r = __builtin_fminf(__builtin_fminf(r2, r3), t_r);
    max_d = __builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(d2), __builtin_fabsf(d3)), __builtin_fabsf(d4));
    const float h6 = h * float(1.0 / 6.0);
    // the last stage's velocities enter the sums as fused products, as in kerr_rk4_step_impl
    const f32x2 sum_rt = pk_fma((f32x2){g.Delta, t_p.y}, (f32x2){g.iSpr, g.iS}, a_rt);
    const f32x2 o_rt = pk_fma(pk_bc(h6), sum_rt, yrt);
    const f32x2 o_p = pk_fma(pk_bc(h6), a_p + g.dp, yp);
    State5<float> o;
    o.r = o_rt.x; o.th = o_rt.y;
    {
        float sn, cn;
        M<float>::sincos(o.th, sn, cn);
        asm volatile("" : "+v"(sn), "+v"(cn)); // (pins the evaluation to this step: the compiler would sink it to its use, the head of the next one)
        cs_out = (f32x2){cn, sn};
    }
    o.ph = __builtin_fmaf(h6, __builtin_fmaf(g.iS, g.u, a_ph), y.ph);
    o.pr = o_p.x; o.pth = o_p.y;
    return o;
}

// What a streak carries from one step to the next besides the state: nothing, or (packed step) [cos, sin] of the polar angle
template <typename T, bool LONE> struct StepCarry {
    __device__ __forceinline__ void init(T) {}
};
template <> struct StepCarry<float, true> {
    f32x2 cs;
    __device__ __forceinline__ void init(float th)
    {
        float s, c;
        M<float>::sincos(th, s, c);
        cs = (f32x2){c, s};
    }
};

// LONE = true: the packed form where there is one (float32); the scalar step otherwise
template <typename T, bool LONE>
__device__ __forceinline__ State5<T> kerr_rk4_step_fast_for(const KerrConsts<T> &k, const RayConsts<T> &rc, const State5<T> &y,
                                                            const StepCarry<T, LONE> &cin, T h, T &min_r, T &max_d,
                                                            StepCarry<T, LONE> &cout)
{
    if constexpr (LONE && sizeof(T) == 4) return kerr_rk4_step_fast_pk(k, rc, y, cin.cs, h, min_r, max_d, cout.cs);
    else return kerr_rk4_step_fast(k, rc, y, h, min_r, max_d);
}

// Step-size rule of the reference's fixed-step RK4 tracer (metrics.py:597-611): h_base, capped in
// three radius bands around the capture radius (tighter caps on axis-refine rays) and by the
// remaining affine range.  Rays spend most of their steps far outside the outermost band
// (r >= 4 r_capture), and a wavefront is an 8x8 pixel tile, so whether ANY lane is inside is decided
// per wave and the band cascade is skipped otherwise.
template <typename T>
__device__ __forceinline__ T kerr_rk4_h(const KerrConsts<T> &k, const RayConsts<T> &rc, T r, T remaining)
{
    T h = rc.hb;
    if (wave_any(r < k.rc4)) {
        T c4 = rc.refine ? T(0.20) : T(0.25), c2 = rc.refine ? T(0.08) : T(0.10), c12 = rc.refine ? T(0.03) : T(0.05);
        h = (r < k.rc4) ? M<T>::min(h, c4) : h;
        h = (r < k.rc2) ? M<T>::min(h, c2) : h;
        h = (r < k.rc12) ? M<T>::min(h, c12) : h;
    }
    return M<T>::min(h, remaining);
}

// Event codes carried from the integrate kernel to the epilogue.
enum : int { EV_MAXRANGE = 2, EV_ESCAPED = 1, EV_CAPTURED = -1, EV_INVALID = 0, EV_PAD = 3, EV_RUNNING = 4 };

// Everything one ray carries between steps (registers).
template <typename T> struct RayState {
    State5<T> y;
    T lam;       // affine parameter so far
    T h_retry;   // > 0: retry the step with this (halved) h, metrics.py:615-626
    uint32_t steps;
};

template <typename T> __device__ __forceinline__ void ray_start(const KerrConsts<T> &k, RayState<T> &s, T p_r, T p_th)
{
    s.y.r = k.r_obs; s.y.th = k.theta_obs; s.y.ph = T(0); s.y.pr = p_r; s.y.pth = p_th;
    s.lam = T(0); s.h_retry = T(0); s.steps = 0;
}

// A streak of ordinary far-field steps.  While EVERY active lane of the wave is outside 4 r_capture with
// nothing to retry, the tracer's loop body (metrics.py:596-655) reduces to: h = min(h_base, remaining), one
// RK4 step, accept.  This loop does exactly that with ONE wave-uniform branch per step -- the predicate
// `good` says that the step was ordinary and that the lane is still in the far field afterwards -- where
// the general iteration below (kerr_rk4_advance) spends four.  Bulk waves take ~60 % of their steps here;
// for a wave hosting one of the few very long rays, alone on its SIMD at the end of the launch, a taken
// branch costs ~80 cycles, so this is what sets the length of the launch's tail.
// A lane for which `good` fails keeps its state (the general iteration redoes that step); lanes for which
// it holds keep the step -- lanes are independent, they need not stay in lockstep.  Same arithmetic as the
// general iteration (kerr_rk4_step_fast, same h), so which of the two paths took a step does not matter.
// Returns the number of loop iterations (wave-uniform).  LONE: the wave is alone on its SIMD (packed step, above).
template <typename T, bool LONE = false>
__device__ __forceinline__ uint32_t kerr_rk4_streak(const KerrConsts<T> &k, const RayConsts<T> &rc, RayState<T> &s,
                                                    uint32_t max_steps)
{
    if (wave_any(!((s.y.r >= k.rc4) & (s.h_retry == T(0))))) return 0;
    uint32_t done = 0;
    // One attempt: from state `from` at affine parameter lam into `to`; true if it was an ordinary far-field step.
    // (predicates are combined with & and |, not && and ||: short-circuit evaluation would turn them into branches)
    // The streak only takes FULL base steps: h = h_base needs remaining >= h_base, i.e. lam <= lambda_max - h_base
    // (the tracer's h = min(h_base, remaining) is then h_base exactly).  A ray within one base step of the range end
    // fails the test and takes its last, shorter step in the general iteration.  With h invariant the step's h / 2 and
    // h / 6, the subtraction and the min leave the loop.
    const T lam_limit = k.lambda_max - rc.hb;
    auto attempt = [&](const State5<T> &from, const StepCarry<T, LONE> &cfrom, T lam, State5<T> &to, StepCarry<T, LONE> &cto, T &h) -> bool {
        h = rc.hb;
        T min_r, max_d;
        to = kerr_rk4_step_fast_for<T, LONE>(k, rc, from, cfrom, h, min_r, max_d, cto);
        T mag = M<T>::abs(to.r) + M<T>::abs(to.th) + M<T>::abs(to.ph) + M<T>::abs(to.pr) + M<T>::abs(to.pth);
        ++done;
        return (lam <= lam_limit) & M<T>::finite(mag) & (to.r >= k.rc4) & (to.r < k.r_escape) & (min_r > k.r_cut) &
               !(max_d > T(0.25));
    };
    // The state ping-pongs between two register sets (A = s.y, B) so that accepting an attempt costs no copies:
    // the loop body is two attempts, A -> B and B -> A, each followed by the one wave-uniform test.  On leaving,
    // a lane keeps the attempted state if its own predicate held, its previous state otherwise.
    State5<T> b;
    StepCarry<T, LONE> ca, cb;
    ca.init(s.y.th);
    T h;
    for (;;) {
        bool good = attempt(s.y, ca, s.lam, b, cb, h);
        if (!wave_all(good) | (done >= max_steps)) {
            s.y.r = good ? b.r : s.y.r; s.y.th = good ? b.th : s.y.th; s.y.ph = good ? b.ph : s.y.ph;
            s.y.pr = good ? b.pr : s.y.pr; s.y.pth = good ? b.pth : s.y.pth;
            s.lam = good ? s.lam + h : s.lam;
            s.steps += good ? 1u : 0u;
            break;
        }
        s.lam += h;
        ++s.steps;
        good = attempt(b, cb, s.lam, s.y, ca, h);
        if (!wave_all(good) | (done >= max_steps)) {
            s.y.r = good ? s.y.r : b.r; s.y.th = good ? s.y.th : b.th; s.y.ph = good ? s.y.ph : b.ph;
            s.y.pr = good ? s.y.pr : b.pr; s.y.pth = good ? s.y.pth : b.pth;
            s.lam = good ? s.lam + h : s.lam;
            s.steps += good ? 1u : 0u;
            break;
        }
        s.lam += h;
        ++s.steps;
    }
    return done;
}

// One iteration of the reference's fixed-step RK4 tracer (metrics.py:596-655): choose h, take an RK4
// step, halve-and-retry on a non-finite result, stop on the capture / escape crossing with the
// reference's linear interpolation.  Returns EV_RUNNING or the terminating event.  Both schedules of
// the integrate kernel call exactly this function, so they produce bit-identical results.
//
// Structure: the arithmetic is straight-line; whether every active lane took an ordinary step (no
// retry, no crossing, range not exhausted) is decided ONCE per wave with a ballot, so the hot path
// is one scalar branch instead of a nest of exec-mask regions; the per-lane logic of the rare cases
// sits behind it.
template <typename T>
__device__ __forceinline__ int kerr_rk4_advance(const KerrConsts<T> &k, const RayConsts<T> &rc, RayState<T> &s)
{
    T remaining = k.lambda_max - s.lam;
    T h = kerr_rk4_h(k, rc, s.y.r, remaining);
    if (__builtin_expect(wave_any(s.h_retry > T(0)), 0)) h = (s.h_retry > T(0)) ? s.h_retry : h;
    bool live = remaining > T(0); // then h > 0 too: every candidate for h is positive
    // (a lane whose range is exhausted still computes the step -- at most once per ray -- and drops it)
    T min_r, max_d;
    State5<T> n = kerr_rk4_step_fast(k, rc, s.y, h, min_r, max_d);
    // the branch-free step is wrong for a lane one of whose stages sat at r <= r_cut or turned by more than
    // 0.25 rad (both rare): such a lane takes the checked step below
    // (& and |, not && and ||: short-circuit evaluation would turn these predicates into branches)
    bool redo = (min_r <= k.r_cut) | (max_d > T(0.25));
    // all five components finite <=> the sum of their magnitudes is (NaN and inf both propagate)
    T mag = M<T>::abs(n.r) + M<T>::abs(n.th) + M<T>::abs(n.ph) + M<T>::abs(n.pr) + M<T>::abs(n.pth);
    bool ok = M<T>::finite(mag) & (n.r > T(0));
    // strictly between the capture and the escape radius: neither crossing test of the reference fires
    bool between = (n.r > k.r_capture) & (n.r < k.r_escape);
    bool plain = live & ok & between & !redo;
    if (__builtin_expect(!wave_any(!plain), 1)) {
        s.y = n;
        s.lam += h;
        s.h_retry = T(0);
        ++s.steps;
        return EV_RUNNING;
    }
    if (wave_any(redo)) {
        // per lane: only a flagged lane takes the checked result, so nobody's numbers depend on a neighbour
        State5<T> e = kerr_rk4_step(k, rc, s.y, h);
        n.r = redo ? e.r : n.r; n.th = redo ? e.th : n.th; n.ph = redo ? e.ph : n.ph;
        n.pr = redo ? e.pr : n.pr; n.pth = redo ? e.pth : n.pth;
        mag = M<T>::abs(n.r) + M<T>::abs(n.th) + M<T>::abs(n.ph) + M<T>::abs(n.pr) + M<T>::abs(n.pth);
        ok = M<T>::finite(mag) && n.r > T(0);
    }
    if (!live) return EV_MAXRANGE;
    ++s.steps;
    if (!ok) {
        T h_floor = M<T>::min(rc.refine ? T(0.01) : T(0.02), rc.hb); // metrics.py:594
        if (h <= h_floor) return EV_INVALID;
        s.h_retry = T(0.5) * h;
        return EV_RUNNING;
    }
    s.h_retry = T(0);
    bool cap_first = n.r <= k.r_capture && s.y.r > k.r_capture; // the reference tests capture first
    bool esc = !cap_first && n.r >= k.r_escape && s.y.r < k.r_escape;
    if (cap_first || esc) {
        T target = cap_first ? k.r_capture : k.r_escape;
        T denom = n.r - s.y.r;
        T frac = (denom == T(0)) ? T(1) : (target - s.y.r) / denom;
        frac = M<T>::min(M<T>::max(frac, T(0)), T(1));
        s.y.r = M<T>::fma(frac, n.r - s.y.r, s.y.r);
        s.y.th = M<T>::fma(frac, n.th - s.y.th, s.y.th);
        s.y.ph = M<T>::fma(frac, n.ph - s.y.ph, s.y.ph);
        s.y.pr = M<T>::fma(frac, n.pr - s.y.pr, s.y.pr);
        s.y.pth = M<T>::fma(frac, n.pth - s.y.pth, s.y.pth);
        return cap_first ? EV_CAPTURED : EV_ESCAPED;
    }
    s.y = n;
    s.lam += h;
    return EV_RUNNING;
}

// ---------------------------------------------------------------------------------------
// Schwarzschild orbit equation u'' = -u + 3 M u^2, u = 1/r (metrics.py:44-117)
// ---------------------------------------------------------------------------------------
template <typename T> struct SchwConsts {
    T M, three_M;
    T u0;         // 1 / r_obs
    T u_capture;  // 1 / (1.01 R_S)
    T u_escape;   // 1 / (2 r_obs)
    T h_max, phi_max;
    uint32_t n_full; // number of full h_max steps that fit below phi_max
    T h_last;        // remaining partial step (0 if none)
};

// Returns the event code; u, w are the final values, phi_last the part of the last step used
// (phi_f = steps_before * h + phi_last is rebuilt in float64 by the epilogue).
template <typename T>
__device__ __forceinline__ int schw_trace(const SchwConsts<T> &k, T &u, T &w, uint32_t &steps, T &phi_last)
{
    int ev = EV_MAXRANGE;
    steps = 0;
    phi_last = T(0);
    uint32_t n_total = k.n_full + (k.h_last > T(0) ? 1u : 0u);
    for (uint32_t i = 0; i < n_total; ++i) {
        T h = (i < k.n_full) ? k.h_max : k.h_last;
        T hh = T(0.5) * h;
        T k1u = w, k1w = M<T>::fma(k.three_M * u, u, -u);
        T tu = M<T>::fma(hh, k1u, u), tw = M<T>::fma(hh, k1w, w);
        T k2u = tw, k2w = M<T>::fma(k.three_M * tu, tu, -tu);
        tu = M<T>::fma(hh, k2u, u); tw = M<T>::fma(hh, k2w, w);
        T k3u = tw, k3w = M<T>::fma(k.three_M * tu, tu, -tu);
        tu = M<T>::fma(h, k3u, u); tw = M<T>::fma(h, k3w, w);
        T k4u = tw, k4w = M<T>::fma(k.three_M * tu, tu, -tu);
        T h6 = h * T(1.0 / 6.0);
        T un = M<T>::fma(h6, k1u + T(2) * (k2u + k3u) + k4u, u);
        T wn = M<T>::fma(h6, k1w + T(2) * (k2w + k3w) + k4w, w);
        bool cap = u < k.u_capture && un >= k.u_capture;
        bool esc = !cap && u > k.u_escape && un <= k.u_escape;
        if (cap || esc) {
            T target = cap ? k.u_capture : k.u_escape;
            T denom = un - u;
            T frac = (denom == T(0)) ? T(1) : (target - u) / denom;
            frac = M<T>::min(M<T>::max(frac, T(0)), T(1));
            phi_last = frac * h;
            w = M<T>::fma(frac, wn - w, w);
            u = target;
            ev = cap ? EV_CAPTURED : EV_ESCAPED;
            break;
        }
        u = un; w = wn;
        ++steps;
    }
    return ev;
}

} // namespace lt
