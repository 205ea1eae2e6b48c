// lt_dp45.hpp -- the reference's production Kerr integrator: adaptive Dormand-Prince 4(5) with FSAL
// (metrics.py:419-567), one step ATTEMPT per call so that both schedules of the integrate kernel
// can drive it.  Float64 only: the reference tolerances (rtol 1e-6 / 1e-8 on axis-refine rays) are at
// or below float32 resolution (in float32 the step controller thrashes).
//
// Also the integrator "policies" the integrate kernels are templated on (Rk4<T>, Dp45<T>).
#pragma once
#include "lt_device.hpp"

namespace lt {

// ---------------------------------------------------------------------------------------------
// fixed-step RK4 as an integrator policy
// ---------------------------------------------------------------------------------------------
template <typename T> struct Rk4 {
    using State = RayState<T>;
    static constexpr int EVALS_FIXED = 0, EVALS_PER_STEP = 4;
    static constexpr int MIN_WAVES_PER_SIMD = 1; // no constraint: float32 fits 5, float64 2
    static constexpr bool GHOST_LANES = true;    // fixed step: a few rays take ~20x the mean step count (k_kerr_direct)
    static __device__ __forceinline__ void start(const KerrConsts<T> &k, const RayConsts<T> &, State &s, T p_r, T p_th)
    {
        ray_start(k, s, p_r, p_th);
    }
    static __device__ __forceinline__ int advance(const KerrConsts<T> &k, const RayConsts<T> &rc, State &s)
    {
        return kerr_rk4_advance(k, rc, s);
    }
    // up to max_steps ordinary far-field steps for the whole wave at one branch per step; returns how many
    static __device__ __forceinline__ uint32_t streak(const KerrConsts<T> &k, const RayConsts<T> &rc, State &s, uint32_t max_steps)
    {
        return kerr_rk4_streak(k, rc, s, max_steps);
    }
    // the same for a wavefront that is alone on its SIMD (ghost-lane phase of the integrate kernels)
    static __device__ __forceinline__ uint32_t streak_lone(const KerrConsts<T> &k, const RayConsts<T> &rc, State &s, uint32_t max_steps)
    {
        return kerr_rk4_streak<T, true>(k, rc, s, max_steps);
    }
};

// ---------------------------------------------------------------------------------------------
// Dormand-Prince 4(5)
// ---------------------------------------------------------------------------------------------
// z^(-1/10) for 1e-30 < z < 1e30 in float, without the transcendental unit (one v_exp / v_log costs
// tens of issue slots in an FMA-dense stream): exponent bit trick (3 % off) + three
// Newton steps y <- y (1 + (1 - z y^10) / 10), relative error ~2e-7.  It only scales the next step
// size h, where 1e-7 is far below what the integration can notice.
__device__ __forceinline__ float pow_minus_tenth(float z)
{
    float y = __uint_as_float((uint32_t)(1171454846.0f - 0.1f * (float)__float_as_uint(z)));
#pragma unroll
    for (int it = 0; it < 3; ++it) {
        float y2 = y * y, y4 = y2 * y2, y5 = y4 * y, y10 = y5 * y5;
        y = __builtin_fmaf(y, 0.1f * __builtin_fmaf(-z, y10, 1.0f), y);
    }
    return y;
}

template <typename T> struct Dp45State {
    State5<T> y;
    T k1[5];       // FSAL: derivative at y
    T s0, c0;      // sin / cos of y.th, carried like k1 (the stages rotate it: sincos_shift)
    T lam, h;
    uint32_t steps; // step attempts (accepted + rejected), metrics.py:454
};

// EXACT_CTRL = false: the step-size controller (error-scale reciprocal, err^-0.2) runs in float32 without the
// transcendental unit -- step sizes follow the reference's to ~1e-7 relative, and the few accept / reject
// decisions that sit within 1e-7 of err_norm = 1 can differ from the reference's (RHS-evaluation counts equal on
// >= 99.9 % of rays, final_alpha to a few 1e-9).  EXACT_CTRL = true (LT_INTEGRATOR_DP45_EXACT): the controller's
// operations as the reference writes them (metrics.py:506-522, :560-564) in float64 -- error over scale, sqrt,
// err_norm ** (-0.2) -- with the division as e * rcp(scale) (~1 ulp), / 5 as * 0.2 and the power as pow_m02 (~1 ulp):
// a decision can differ from the reference's only when err_norm is within a few ulp of 1 (measured: on no golden ray).
template <typename T, bool EXACT_CTRL = false> struct Dp45 {
    using State = Dp45State<T>;
    static constexpr int EVALS_FIXED = 1, EVALS_PER_STEP = 6;
    // hold the register allocation at 256 (it sits just above); the exact controller fits there too since its
    // err_norm ** -0.2 is pow_m02 (lt_device.hpp) and no longer the float64 library pow
    static constexpr int MIN_WAVES_PER_SIMD = 2;
    static constexpr bool GHOST_LANES = false;   // adaptive steps: no ray is long enough to be alone on the chip

    static __device__ __forceinline__ void start(const KerrConsts<T> &k, const RayConsts<T> &rc, State &s, T p_r, T p_th)
    {
        s.y.r = k.r_obs; s.y.th = k.theta_obs; s.y.ph = T(0); s.y.pr = p_r; s.y.pth = p_th;
        M<T>::sincos(s.y.th, s.s0, s.c0);
        kerr_rhs_sc(k, rc, s.y.r, s.s0, s.c0, s.y.pr, s.y.pth, s.k1[0], s.k1[1], s.k1[2], s.k1[3], s.k1[4]); // metrics.py:446
        s.lam = T(0);
        s.h = M<T>::max(T(1), T(0.01) * k.r_obs);           // metrics.py:449
        s.steps = 0;
    }

    static __device__ __forceinline__ uint32_t streak(const KerrConsts<T> &, const RayConsts<T> &, State &, uint32_t) { return 0; }
    static __device__ __forceinline__ uint32_t streak_lone(const KerrConsts<T> &, const RayConsts<T> &, State &, uint32_t) { return 0; }

    // right-hand side at stage state y, whose polar angle is th0 + dth with (s0, c0) = sincos(th0).
    // CHECKED = false: without the in-line r <= r_cut and |dth| > 0.25 handling (see attempt()).
    template <bool CHECKED>
    static __device__ __forceinline__ void rhs5(const KerrConsts<T> &k, const RayConsts<T> &rc, const T *y, T th0, T s0,
                                                T c0, T dth, T *d, T &s, T &c)
    {
        sincos_shift<T, CHECKED>(th0, s0, c0, dth, s, c);
        kerr_rhs_sc<T, CHECKED>(k, rc, y[0], s, c, y[3], y[4], d[0], d[1], d[2], d[3], d[4]);
    }

    // The arithmetic of one step attempt: six stages, the candidate next state nxt with its derivative k7 (FSAL)
    // and trigonometry (sn, cn), and the squared error norm.  The unchecked variant has no branches; it reports
    // the smallest stage radius and the largest stage angle offset so that the caller can tell whether the
    // checked variant (fourteen wave-uniform tests per attempt) was needed for this lane.
    template <bool CHECKED>
    static __device__ __forceinline__ void attempt(const KerrConsts<T> &k, const RayConsts<T> &rc, const State &s, T h,
                                                   T *nxt, T *k7, T &sn, T &cn, T &err_sq, T &min_r, T &max_d)
    {
        const T A21 = T(1.0 / 5.0), A31 = T(3.0 / 40.0), A32 = T(9.0 / 40.0), A41 = T(44.0 / 45.0), A42 = T(-56.0 / 15.0),
                A43 = T(32.0 / 9.0), A51 = T(19372.0 / 6561.0), A52 = T(-25360.0 / 2187.0), A53 = T(64448.0 / 6561.0),
                A54 = T(-212.0 / 729.0), A61 = T(9017.0 / 3168.0), A62 = T(-355.0 / 33.0), A63 = T(46732.0 / 5247.0),
                A64 = T(49.0 / 176.0), A65 = T(-5103.0 / 18656.0);
        const T B1 = T(35.0 / 384.0), B3 = T(500.0 / 1113.0), B4 = T(125.0 / 192.0), B5 = T(-2187.0 / 6784.0), B6 = T(11.0 / 84.0);
        const T E1 = T(71.0 / 57600.0), E3 = T(-71.0 / 16695.0), E4 = T(71.0 / 1920.0), E5 = T(-17253.0 / 339200.0),
                E6 = T(22.0 / 525.0), E7 = T(-1.0 / 40.0);
        const T y[5] = {s.y.r, s.y.th, s.y.ph, s.y.pr, s.y.pth};
        T k2[5], k3[5], k4[5], k5[5], k6[5], tmp[5];
        const T *k1 = s.k1;
        T ss, cc;
        const T th0 = y[1], s0 = s.s0, c0 = s.c0;
        T d2, d3, d4, d5, d6, d7, r2, r3, r4, r5, r6;
#pragma unroll
        for (int i = 0; i < 5; ++i) tmp[i] = y[i] + h * A21 * k1[i];
        d2 = tmp[1] - th0; r2 = tmp[0];
        rhs5<CHECKED>(k, rc, tmp, th0, s0, c0, d2, k2, ss, cc);
#pragma unroll
        for (int i = 0; i < 5; ++i) tmp[i] = y[i] + h * (A31 * k1[i] + A32 * k2[i]);
        d3 = tmp[1] - th0; r3 = tmp[0];
        rhs5<CHECKED>(k, rc, tmp, th0, s0, c0, d3, k3, ss, cc);
#pragma unroll
        for (int i = 0; i < 5; ++i) tmp[i] = y[i] + h * (A41 * k1[i] + A42 * k2[i] + A43 * k3[i]);
        d4 = tmp[1] - th0; r4 = tmp[0];
        rhs5<CHECKED>(k, rc, tmp, th0, s0, c0, d4, k4, ss, cc);
#pragma unroll
        for (int i = 0; i < 5; ++i) tmp[i] = y[i] + h * (A51 * k1[i] + A52 * k2[i] + A53 * k3[i] + A54 * k4[i]);
        d5 = tmp[1] - th0; r5 = tmp[0];
        rhs5<CHECKED>(k, rc, tmp, th0, s0, c0, d5, k5, ss, cc);
#pragma unroll
        for (int i = 0; i < 5; ++i)
            tmp[i] = y[i] + h * (A61 * k1[i] + A62 * k2[i] + A63 * k3[i] + A64 * k4[i] + A65 * k5[i]);
        d6 = tmp[1] - th0; r6 = tmp[0];
        rhs5<CHECKED>(k, rc, tmp, th0, s0, c0, d6, k6, ss, cc);
#pragma unroll
        for (int i = 0; i < 5; ++i) nxt[i] = y[i] + h * (B1 * k1[i] + B3 * k3[i] + B4 * k4[i] + B5 * k5[i] + B6 * k6[i]);
        d7 = nxt[1] - th0;
        rhs5<CHECKED>(k, rc, nxt, th0, s0, c0, d7, k7, sn, cn);
        if (!CHECKED) { // (min / max ignore a NaN stage value; it makes nxt non-finite in either variant)
            min_r = M<T>::min(M<T>::min(M<T>::min(r2, r3), M<T>::min(r4, r5)), M<T>::min(r6, nxt[0]));
            max_d = M<T>::max(M<T>::max(M<T>::max(M<T>::abs(d2), M<T>::abs(d3)), M<T>::max(M<T>::abs(d4), M<T>::abs(d5))),
                              M<T>::max(M<T>::abs(d6), M<T>::abs(d7)));
        }
        const T atol = rc.refine ? T(1e-10) : T(1e-8), rtol = rc.refine ? T(1e-8) : T(1e-6); // metrics.py:431-432
        err_sq = T(0);
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            T ei = h * (E1 * k1[i] + E3 * k3[i] + E4 * k4[i] + E5 * k5[i] + E6 * k6[i] + E7 * k7[i]);
            T sc = atol + rtol * M<T>::max(M<T>::abs(y[i]), M<T>::abs(nxt[i]));
            // ei / sc with a float reciprocal (relative error 1e-7): the error norm only gates accept / reject
            // and scales h; a float64 division is ~12 instructions, five of them per attempt.  The exact controller
            // takes the float64 reciprocal (seed + two Newton steps, within an ulp).
            T q = EXACT_CTRL ? ei * (T)M<double>::rcp_pos((double)sc) : ei * (T)M<float>::rcp_pos((float)sc);
            err_sq += q * q;
        }
    }

    // One attempt of the loop body metrics.py:454-564.
    static __device__ __forceinline__ int advance(const KerrConsts<T> &k, const RayConsts<T> &rc, State &s)
    {
        const T h_min = T(1e-12);
        if (s.steps >= 200000u || !(s.lam < k.lambda_max)) return EV_MAXRANGE;
        T remaining = k.lambda_max - s.lam;
        if (s.h > remaining) s.h = remaining;
        if (!(s.h > T(0))) return EV_MAXRANGE;
        ++s.steps;
        const T h = s.h;
        const T y[5] = {s.y.r, s.y.th, s.y.ph, s.y.pr, s.y.pth};
        T nxt[5], k7[5], sn, cn, err_sq, min_r, max_d;
        attempt<false>(k, rc, s, h, nxt, k7, sn, cn, err_sq, min_r, max_d);
        // one test per attempt instead of fourteen: a lane one of whose stages sat at r <= r_cut (the last attempts
        // before capture) or turned by more than 0.25 rad takes the checked arithmetic -- that lane only, so
        // nobody's numbers depend on a neighbour
        const bool redo = (min_r <= k.r_cut) | (max_d > T(0.25));
        if (__builtin_expect(wave_any(redo), 0)) {
            T nx2[5], k72[5], sn2, cn2, e2, u1, u2;
            attempt<true>(k, rc, s, h, nx2, k72, sn2, cn2, e2, u1, u2);
#pragma unroll
            for (int i = 0; i < 5; ++i) { nxt[i] = redo ? nx2[i] : nxt[i]; k7[i] = redo ? k72[i] : k7[i]; }
            sn = redo ? sn2 : sn; cn = redo ? cn2 : cn; err_sq = redo ? e2 : err_sq;
        }

        T mag = M<T>::abs(nxt[0]) + M<T>::abs(nxt[1]) + M<T>::abs(nxt[2]) + M<T>::abs(nxt[3]) + M<T>::abs(nxt[4]);
        if (!(M<T>::finite(mag) && nxt[0] > T(0))) { // metrics.py:500-504
            s.h *= T(0.25);
            return s.h < h_min ? EV_INVALID : EV_RUNNING;
        }
        // err_norm = sqrt(err_sq / 5); err_norm^(-0.2) = (err_sq / 5)^(-0.1)
        T grow;
        bool reject, tiny;
        if (EXACT_CTRL) {
            const double err_norm = __builtin_sqrt((double)err_sq * 0.2);          // metrics.py:514 (err_sq / 5 to an ulp)
            grow = (T)(0.9 * pow_m02(err_norm));                                    // metrics.py:518, :564 (err_norm ** -0.2)
            reject = err_norm > 1.0;
            tiny = err_norm < 1e-10;
        } else {
            grow = T(0.9) * (T)pow_minus_tenth((float)(err_sq * T(0.2)));
            reject = err_sq > T(5);
            tiny = err_sq < T(5e-20);
        }
        if (reject) { // err_norm > 1: reject, metrics.py:516-522
            s.h *= M<T>::max(T(0.2), grow);
            return s.h < h_min ? EV_INVALID : EV_RUNNING;
        }
        // accept
        bool cap = y[0] > k.r_capture && nxt[0] <= k.r_capture;
        bool esc = !cap && y[0] < k.r_escape && nxt[0] >= k.r_escape;
        if (cap || esc) {
            T target = cap ? k.r_capture : k.r_escape;
            T denom = nxt[0] - y[0];
            T frac = (denom == T(0)) ? T(1) : (target - y[0]) / denom;
            frac = M<T>::min(M<T>::max(frac, T(0)), T(1));
            s.y.r = y[0] + frac * (nxt[0] - y[0]);
            s.y.th = y[1] + frac * (nxt[1] - y[1]);
            s.y.ph = y[2] + frac * (nxt[2] - y[2]);
            s.y.pr = y[3] + frac * (nxt[3] - y[3]);
            s.y.pth = y[4] + frac * (nxt[4] - y[4]);
            return cap ? EV_CAPTURED : EV_ESCAPED;
        }
        s.y.r = nxt[0]; s.y.th = nxt[1]; s.y.ph = nxt[2]; s.y.pr = nxt[3]; s.y.pth = nxt[4];
#pragma unroll
        for (int i = 0; i < 5; ++i) s.k1[i] = k7[i];
        s.s0 = sn; s.c0 = cn; // FSAL for the trigonometry too
        s.lam += h;
        s.h = tiny ? h * T(5) : h * M<T>::min(T(5), grow); // err_norm < 1e-10, metrics.py:561-564
        return EV_RUNNING;
    }
};

} // namespace lt
