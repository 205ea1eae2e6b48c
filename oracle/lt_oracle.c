/*
 * oracle/lt_oracle.c  --  TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * A plain-C, CPU restatement of the per-pixel backward light-ray integrator of
 * dhg14n9/Light-path-tracer (reference: /root/reference, pure Python/numba).
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library; the product (light-path-tracer_amd/) never links, imports
 * or falls back to it.
 *
 * Parity status: PINNED.  Every function below is checked in tests/ against
 * golden vectors produced by importing the reference itself in the build
 * container (tests/golden/make_golden.py, fixtures F1..F9).
 *
 * The arithmetic follows the reference operation by operation (same order,
 * same constants, no FMA contraction: build with -ffp-contract=off), so in
 * float64 it agrees with the reference to libm last-bit effects.
 *
 * A float32 build of the same file (-DLTO_F32, symbols suffixed _f32) exists
 * only to attribute GPU-vs-reference differences to float32 arithmetic.
 *
 * Each function cites the reference file:line it restates.
 */
#define _GNU_SOURCE
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#ifdef LTO_F32
typedef float real;
#define R(x) x##f
#define LTO_NAME(x) x##_f32
#define M_SIN sinf
#define M_COS cosf
#define M_SQRT sqrtf
#define M_ACOS acosf
#define M_ATAN2 atan2f
#define M_FABS fabsf
#define M_FLOOR floorf
#define M_FMOD fmodf
#define M_POW powf
#define M_COPYSIGN copysignf
#else
typedef double real;
#define R(x) x
#define LTO_NAME(x) x
#define M_SIN sin
#define M_COS cos
#define M_SQRT sqrt
#define M_ACOS acos
#define M_ATAN2 atan2
#define M_FABS fabs
#define M_FLOOR floor
#define M_FMOD fmod
#define M_POW pow
#define M_COPYSIGN copysign
#endif

#define LTO_PI R(3.141592653589793)
#define LTO_NAN ((real)NAN)

/* per-thread RHS evaluation counter (fixture F3 records rhs_evals) */
static _Thread_local uint32_t lto_evals;

/* metrics.py:35-41 */
static real clip_scalar(real x, real lo, real hi)
{
    if (x < lo) return lo;
    if (x > hi) return hi;
    return x;
}

/* Python/numpy float floor-division a // b for b > 0 (CPython float_divmod /
 * numpy npy_divmod), used at metrics.py:133 and :372 as int(abs(phi)//pi). */
static real floor_div(real a, real b)
{
    real mod = M_FMOD(a, b);
    real div = (a - mod) / b;
    if (mod != R(0.0)) {
        if ((b < R(0.0)) != (mod < R(0.0))) { mod += b; div -= R(1.0); }
    }
    real fl;
    if (div != R(0.0)) {
        fl = M_FLOOR(div);
        if (div - fl > R(0.5)) fl += R(1.0);
    } else {
        fl = M_COPYSIGN(R(0.0), a / b);
    }
    return fl;
}

static int all_finite5(const real *x) /* metrics.py:326-331 */
{
    for (int i = 0; i < 5; ++i)
        if (!isfinite(x[i])) return 0;
    return 1;
}

/* ------------------------------------------------------------------------ */
/* Schwarzschild orbit-equation tracer                                       */
/* ------------------------------------------------------------------------ */

/* metrics.py:44-46 */
static void schw_rhs(real u, real w, real M, real *du, real *dw)
{
    lto_evals++;
    *du = w;
    *dw = -u + R(3.0) * M * u * u;
}

/* metrics.py:49-117; status 1 escaped, -1 captured, 0 invalid, 2 max-range */
static int schw_trace_orbit(real M, real R_S, real r_obs, real alpha, real phi_max, real h_max,
                            real *phi_out, real *u_out, real *w_out)
{
    *phi_out = LTO_NAN; *u_out = R(0.0); *w_out = R(0.0);
    real f0 = R(1.0) - R_S / r_obs;
    if (f0 <= R(0.0)) return 0;
    real b = r_obs * M_SIN(alpha) / M_SQRT(f0);
    if (b == R(0.0)) return 0;
    real u = R(1.0) / r_obs;
    real w0_sq = R(1.0) / (b * b) - u * u + R(2.0) * M * u * u * u;
    if (w0_sq < R(0.0)) return 0;
    real w = M_SQRT(w0_sq);
    real phi = R(0.0);
    real u_capture = R(1.0) / (R_S * R(1.01));
    real u_escape = R(1.0) / (R(2.0) * r_obs);
    int status = 2;
    while (phi < phi_max) {
        real h = h_max;
        real remaining = phi_max - phi;
        if (remaining < h) h = remaining;
        if (h <= R(0.0)) break;
        real u_prev = u, w_prev = w;
        real k1u, k1w, k2u, k2w, k3u, k3w, k4u, k4w;
        schw_rhs(u_prev, w_prev, M, &k1u, &k1w);
        schw_rhs(u_prev + R(0.5) * h * k1u, w_prev + R(0.5) * h * k1w, M, &k2u, &k2w);
        schw_rhs(u_prev + R(0.5) * h * k2u, w_prev + R(0.5) * h * k2w, M, &k3u, &k3w);
        schw_rhs(u_prev + h * k3u, w_prev + h * k3w, M, &k4u, &k4w);
        u = u_prev + (h / R(6.0)) * (k1u + R(2.0) * k2u + R(2.0) * k3u + k4u);
        w = w_prev + (h / R(6.0)) * (k1w + R(2.0) * k2w + R(2.0) * k3w + k4w);
        real phi_next = phi + h;
        if (u_prev < u_capture && u >= u_capture) {
            real denom = u - u_prev;
            real frac = (denom == R(0.0)) ? R(1.0) : (u_capture - u_prev) / denom;
            frac = clip_scalar(frac, R(0.0), R(1.0));
            phi = phi + frac * h;
            w = w_prev + frac * (w - w_prev);
            u = u_capture;
            status = -1;
            break;
        }
        if (u_prev > u_escape && u <= u_escape) {
            real denom = u - u_prev;
            real frac = (denom == R(0.0)) ? R(1.0) : (u_escape - u_prev) / denom;
            frac = clip_scalar(frac, R(0.0), R(1.0));
            phi = phi + frac * h;
            w = w_prev + frac * (w - w_prev);
            u = u_escape;
            status = 1;
            break;
        }
        phi = phi_next;
    }
    *phi_out = phi; *u_out = u; *w_out = w;
    return status;
}

/* metrics.py:120-145 -> (status, final_alpha, n_half) */
static int schw_trace_ray(real M, real R_S, real r_obs, real alpha, real phi_max, real h_max,
                          real *fa_out, int64_t *nh_out)
{
    real phi_f, u_f, w_f;
    int status = schw_trace_orbit(M, R_S, r_obs, alpha, phi_max, h_max, &phi_f, &u_f, &w_f);
    *fa_out = LTO_NAN; *nh_out = 0;
    if (status == 0) return 0;
    real r_f = R(1.0) / u_f;
    int64_t nh = (int64_t)floor_div(M_FABS(phi_f), LTO_PI);
    *nh_out = nh;
    if (status == -1 || r_f <= R_S * R(1.1)) return -1;
    real dr_dphi = -w_f / (u_f * u_f);
    real sin_phi = M_SIN(phi_f), cos_phi = M_COS(phi_f);
    real heading = M_ATAN2(dr_dphi * sin_phi + r_f * cos_phi,
                           dr_dphi * cos_phi - r_f * sin_phi);
    *fa_out = M_ACOS(clip_scalar(-M_COS(heading), R(-1.0), R(1.0)));
    return 1;
}

/* ------------------------------------------------------------------------ */
/* Kerr                                                                      */
/* ------------------------------------------------------------------------ */

/* metrics.py:148-218; returns ok, fills state5 = [r, theta, phi, p_r, p_theta] */
static int kerr_ic(real M, real a, real r_obs, real alpha, real theta, real theta_obs,
                   real *state, real *p_t_out, real *p_phi_out)
{
    real r = r_obs, th = theta_obs;
    real sin_th = M_SIN(th), cos_th = M_COS(th);
    real sin_th_sq = sin_th * sin_th;
    if (sin_th_sq < R(1e-15)) sin_th_sq = R(1e-15);
    real Sigma = r * r + a * a * cos_th * cos_th;
    real Delta = r * r - R(2.0) * M * r + a * a;
    *p_t_out = R(0.0); *p_phi_out = R(0.0);
    if (Delta <= R(0.0) || Sigma <= R(0.0)) return 0;
    real sin_alpha = M_SIN(alpha);
    real sin_screen = M_SIN(theta), cos_screen = M_COS(theta);
    real E = R(1.0);
    real sqrt_Delta = M_SQRT(Delta), sqrt_Sigma = M_SQRT(Sigma);
    real rho = r * sin_alpha * sqrt_Sigma / sqrt_Delta;
    real alpha_screen = -rho * sin_screen;
    real beta_screen = -rho * cos_screen;
    real xi = -alpha_screen * sin_th;
    real eta = beta_screen * beta_screen + cos_th * cos_th * (alpha_screen * alpha_screen - a * a);
    real L = xi * E;
    real Q = eta * E * E;
    real p_t = -E;
    real p_phi = L;
    real Theta = Q - cos_th * cos_th * (L * L / sin_th_sq - a * a * E * E);
    if (Theta < R(0.0)) Theta = R(0.0);
    real p_th_sign = (cos_screen > R(0.0)) ? R(-1.0) : R(1.0);
    real p_theta = p_th_sign * M_SQRT(Theta);
    real A_val = (r * r + a * a) * (r * r + a * a) - a * a * Delta * sin_th_sq;
    real g_tt_inv = -A_val / (Sigma * Delta);
    real g_tphi_inv = R(-2.0) * M * a * r / (Sigma * Delta);
    real g_rr_inv = Delta / Sigma;
    real g_thth_inv = R(1.0) / Sigma;
    real g_phiphi_inv = (Delta - a * a * sin_th_sq) / (Sigma * Delta * sin_th_sq);
    real other = (g_tt_inv * p_t * p_t
                  + R(2.0) * g_tphi_inv * p_t * p_phi
                  + g_thth_inv * p_theta * p_theta
                  + g_phiphi_inv * p_phi * p_phi);
    real p_r_sq = -other / g_rr_inv;
    if (p_r_sq < R(0.0)) p_r_sq = R(0.0);
    real p_r = -M_SQRT(p_r_sq);
    state[0] = r; state[1] = th; state[2] = R(0.0); state[3] = p_r; state[4] = p_theta;
    *p_t_out = p_t; *p_phi_out = p_phi;
    return 1;
}

/* metrics.py:221-303 */
static void kerr_rhs(const real *s, real p_t, real p_phi, real M, real a, real r_plus, real *out)
{
    lto_evals++;
    real r = s[0], th = s[1], p_r = s[3], p_th = s[4];
    if (r <= r_plus * R(1.001)) {
        for (int i = 0; i < 5; ++i) out[i] = R(0.0);
        return;
    }
    real sin_th = M_SIN(th), cos_th = M_COS(th);
    real sin_th_sq = sin_th * sin_th;
    if (sin_th_sq < R(1e-15)) sin_th_sq = R(1e-15);
    real Sigma = r * r + a * a * cos_th * cos_th;
    real Delta = r * r - R(2.0) * M * r + a * a;
    real A = (r * r + a * a) * (r * r + a * a) - a * a * Delta * sin_th_sq;

    real g_tphi_inv = R(-2.0) * M * a * r / (Sigma * Delta);
    real g_rr_inv = Delta / Sigma;
    real g_thth_inv = R(1.0) / Sigma;
    real g_phiphi_inv = (Delta - a * a * sin_th_sq) / (Sigma * Delta * sin_th_sq);

    real dr = g_rr_inv * p_r;
    real dth = g_thth_inv * p_th;
    real dphi = g_tphi_inv * p_t + g_phiphi_inv * p_phi;

    real dSigma_dr = R(2.0) * r;
    real dDelta_dr = R(2.0) * r - R(2.0) * M;
    real dA_dr = R(4.0) * r * (r * r + a * a) - a * a * dDelta_dr * sin_th_sq;

    real sigma_delta = Sigma * Delta;
    real sigma_delta_sq = sigma_delta * sigma_delta;
    real dg_tt_inv_dr = (-(dA_dr * sigma_delta
                           - A * (dSigma_dr * Delta + Sigma * dDelta_dr))
                         / sigma_delta_sq);
    real dg_tphi_inv_dr = (-(R(2.0) * M * a * (sigma_delta
                              - r * (dSigma_dr * Delta + Sigma * dDelta_dr)))
                           / sigma_delta_sq);
    real dg_rr_inv_dr = (dDelta_dr * Sigma - Delta * dSigma_dr) / (Sigma * Sigma);
    real dg_thth_inv_dr = -dSigma_dr / (Sigma * Sigma);
    real den_phi_dr = Sigma * Delta * sin_th_sq;
    real dg_phiphi_inv_dr = ((dDelta_dr * den_phi_dr
                              - (Delta - a * a * sin_th_sq)
                              * (dSigma_dr * Delta + Sigma * dDelta_dr) * sin_th_sq)
                             / (den_phi_dr * den_phi_dr));

    real dp_r = R(-0.5) * (dg_tt_inv_dr * p_t * p_t
                           + R(2.0) * dg_tphi_inv_dr * p_t * p_phi
                           + dg_rr_inv_dr * p_r * p_r
                           + dg_thth_inv_dr * p_th * p_th
                           + dg_phiphi_inv_dr * p_phi * p_phi);

    real dSigma_dth = R(-2.0) * a * a * sin_th * cos_th;
    real dA_dth = -a * a * Delta * R(2.0) * sin_th * cos_th;

    real dg_tt_inv_dth = (-(dA_dth * Sigma * Delta - A * dSigma_dth * Delta)
                          / sigma_delta_sq);
    real dg_tphi_inv_dth = R(2.0) * M * a * r * dSigma_dth / (Sigma * Sigma * Delta);
    real dg_rr_inv_dth = -Delta * dSigma_dth / (Sigma * Sigma);
    real dg_thth_inv_dth = -dSigma_dth / (Sigma * Sigma);

    real num = Delta - a * a * sin_th_sq;
    real den = Sigma * Delta * sin_th_sq;
    real dnum_dth = -a * a * R(2.0) * sin_th * cos_th;
    real dden_dth = dSigma_dth * Delta * sin_th_sq + Sigma * Delta * R(2.0) * sin_th * cos_th;
    real dg_phiphi_inv_dth = (dnum_dth * den - num * dden_dth) / (den * den);

    real dp_th = R(-0.5) * (dg_tt_inv_dth * p_t * p_t
                            + R(2.0) * dg_tphi_inv_dth * p_t * p_phi
                            + dg_rr_inv_dth * p_r * p_r
                            + dg_thth_inv_dth * p_th * p_th
                            + dg_phiphi_inv_dth * p_phi * p_phi);

    out[0] = dr; out[1] = dth; out[2] = dphi; out[3] = dp_r; out[4] = dp_th;
}

/* metrics.py:306-323 */
static void rk4_step_kerr(const real *state, real h, real p_t, real p_phi, real M, real a,
                          real r_plus, real *k1, real *k2, real *k3, real *k4, real *tmp,
                          real *out_state)
{
    kerr_rhs(state, p_t, p_phi, M, a, r_plus, k1);
    for (int i = 0; i < 5; ++i) tmp[i] = state[i] + R(0.5) * h * k1[i];
    kerr_rhs(tmp, p_t, p_phi, M, a, r_plus, k2);
    for (int i = 0; i < 5; ++i) tmp[i] = state[i] + R(0.5) * h * k2[i];
    kerr_rhs(tmp, p_t, p_phi, M, a, r_plus, k3);
    for (int i = 0; i < 5; ++i) tmp[i] = state[i] + h * k3[i];
    kerr_rhs(tmp, p_t, p_phi, M, a, r_plus, k4);
    for (int i = 0; i < 5; ++i)
        out_state[i] = state[i] + (h / R(6.0)) * (k1[i] + R(2.0) * k2[i] + R(2.0) * k3[i] + k4[i]);
}

/* metrics.py:363-416 */
static int kerr_extract_angle(const real *state, real p_t, real p_phi, real M, real a,
                              real r_capture, int event_status, real *fa_out, int64_t *nh_out)
{
    real r_f = state[0], th_f = state[1], phi_f = state[2], p_r_f = state[3], p_th_f = state[4];
    int64_t nh = (int64_t)floor_div(M_FABS(phi_f), LTO_PI);
    *fa_out = LTO_NAN; *nh_out = nh;
    if (r_f <= r_capture * R(1.1) || event_status == -1) return -1;
    if (!isfinite(r_f) || !isfinite(th_f) || !isfinite(phi_f)) { *nh_out = 0; return 0; }
    real sin_th = M_SIN(th_f), cos_th = M_COS(th_f);
    real sin_th_sq = sin_th * sin_th;
    if (sin_th_sq < R(1e-15)) sin_th_sq = R(1e-15);
    real Sigma_f = r_f * r_f + a * a * cos_th * cos_th;
    real Delta_f = r_f * r_f - R(2.0) * M * r_f + a * a;
    if (Sigma_f <= R(1e-15) || M_FABS(Delta_f) <= R(1e-15)) return 0;
    real dr_dl = Delta_f / Sigma_f * p_r_f;
    real dth_dl = p_th_f / Sigma_f;
    real dphi_dl = (R(-2.0) * M * a * r_f / (Sigma_f * Delta_f) * p_t
                    + (Delta_f - a * a * sin_th_sq)
                    / (Sigma_f * Delta_f * sin_th_sq) * p_phi);
    real sin_phi = M_SIN(phi_f), cos_phi = M_COS(phi_f);
    real vx = (sin_th * cos_phi * dr_dl
               + r_f * cos_th * cos_phi * dth_dl
               - r_f * sin_th * sin_phi * dphi_dl);
    real vy = (sin_th * sin_phi * dr_dl
               + r_f * cos_th * sin_phi * dth_dl
               + r_f * sin_th * cos_phi * dphi_dl);
    real vz = cos_th * dr_dl - r_f * sin_th * dth_dl;
    if (!isfinite(vx) || !isfinite(vy) || !isfinite(vz)) return 0;
    real v_mag = M_SQRT(vx * vx + vy * vy + vz * vz);
    if (v_mag < R(1e-30)) return 1; /* escaped with NaN angle */
    *fa_out = M_ACOS(clip_scalar(-vx / v_mag, R(-1.0), R(1.0)));
    return 1;
}

/* Dormand-Prince tableau, metrics.py:334-360 */
#define DP_A21 (R(1.0) / R(5.0))
#define DP_A31 (R(3.0) / R(40.0))
#define DP_A32 (R(9.0) / R(40.0))
#define DP_A41 (R(44.0) / R(45.0))
#define DP_A42 (R(-56.0) / R(15.0))
#define DP_A43 (R(32.0) / R(9.0))
#define DP_A51 (R(19372.0) / R(6561.0))
#define DP_A52 (R(-25360.0) / R(2187.0))
#define DP_A53 (R(64448.0) / R(6561.0))
#define DP_A54 (R(-212.0) / R(729.0))
#define DP_A61 (R(9017.0) / R(3168.0))
#define DP_A62 (R(-355.0) / R(33.0))
#define DP_A63 (R(46732.0) / R(5247.0))
#define DP_A64 (R(49.0) / R(176.0))
#define DP_A65 (R(-5103.0) / R(18656.0))
#define DP_B1 (R(35.0) / R(384.0))
#define DP_B3 (R(500.0) / R(1113.0))
#define DP_B4 (R(125.0) / R(192.0))
#define DP_B5 (R(-2187.0) / R(6784.0))
#define DP_B6 (R(11.0) / R(84.0))
#define DP_E1 (R(71.0) / R(57600.0))
#define DP_E3 (R(-71.0) / R(16695.0))
#define DP_E4 (R(71.0) / R(1920.0))
#define DP_E5 (R(-17253.0) / R(339200.0))
#define DP_E6 (R(22.0) / R(525.0))
#define DP_E7 (R(-1.0) / R(40.0))

static real rmax(real a, real b) { return a > b ? a : b; }
static real rmin(real a, real b) { return a < b ? a : b; }

/* metrics.py:419-567, production integrator (DP45, FSAL) */
static int kerr_trace_dp45(real M, real a, real r_plus, real r_obs, real alpha, real theta,
                           real theta_obs, real lambda_max, real h_max, int axis_refine,
                           real *fa_out, int64_t *nh_out)
{
    (void)h_max; /* unused by the reference as well */
    real state[5], p_t, p_phi;
    *fa_out = LTO_NAN; *nh_out = 0;
    if (!kerr_ic(M, a, r_obs, alpha, theta, theta_obs, state, &p_t, &p_phi)) return 0;
    real r_capture = r_plus * R(1.01);
    real r_escape = r_obs * R(2.0);
    real atol = axis_refine ? R(1e-10) : R(1e-8);
    real rtol = axis_refine ? R(1e-8) : R(1e-6);
    real k1[5], k2[5], k3[5], k4[5], k5[5], k6[5], k7[5], tmp[5], next_state[5];
    kerr_rhs(state, p_t, p_phi, M, a, r_plus, k1);
    real lam = R(0.0);
    real h = rmax(R(1.0), R(0.01) * r_obs);
    real h_min = R(1e-12);
    int event_status = 2;
    const int max_steps = 200000;
    for (int step = 0; step < max_steps; ++step) {
        if (lam >= lambda_max) break;
        real remaining = lambda_max - lam;
        if (h > remaining) h = remaining;
        if (h <= R(0.0)) break;
        for (int i = 0; i < 5; ++i) tmp[i] = state[i] + h * DP_A21 * k1[i];
        kerr_rhs(tmp, p_t, p_phi, M, a, r_plus, k2);
        for (int i = 0; i < 5; ++i) tmp[i] = state[i] + h * (DP_A31 * k1[i] + DP_A32 * k2[i]);
        kerr_rhs(tmp, p_t, p_phi, M, a, r_plus, k3);
        for (int i = 0; i < 5; ++i)
            tmp[i] = state[i] + h * (DP_A41 * k1[i] + DP_A42 * k2[i] + DP_A43 * k3[i]);
        kerr_rhs(tmp, p_t, p_phi, M, a, r_plus, k4);
        for (int i = 0; i < 5; ++i)
            tmp[i] = state[i] + h * (DP_A51 * k1[i] + DP_A52 * k2[i]
                                     + DP_A53 * k3[i] + DP_A54 * k4[i]);
        kerr_rhs(tmp, p_t, p_phi, M, a, r_plus, k5);
        for (int i = 0; i < 5; ++i)
            tmp[i] = state[i] + h * (DP_A61 * k1[i] + DP_A62 * k2[i] + DP_A63 * k3[i]
                                     + DP_A64 * k4[i] + DP_A65 * k5[i]);
        kerr_rhs(tmp, p_t, p_phi, M, a, r_plus, k6);
        for (int i = 0; i < 5; ++i)
            next_state[i] = state[i] + h * (DP_B1 * k1[i] + DP_B3 * k3[i] + DP_B4 * k4[i]
                                            + DP_B5 * k5[i] + DP_B6 * k6[i]);
        kerr_rhs(next_state, p_t, p_phi, M, a, r_plus, k7);
        if (!all_finite5(next_state) || next_state[0] <= R(0.0)) {
            h *= R(0.25);
            if (h < h_min) return 0;
            continue;
        }
        real err_sq = R(0.0);
        for (int idx = 0; idx < 5; ++idx) {
            real ei = h * (DP_E1 * k1[idx] + DP_E3 * k3[idx]
                           + DP_E4 * k4[idx] + DP_E5 * k5[idx]
                           + DP_E6 * k6[idx] + DP_E7 * k7[idx]);
            real sc = atol + rtol * rmax(M_FABS(state[idx]), M_FABS(next_state[idx]));
            real q = ei / sc;
            err_sq += q * q;
        }
        real err_norm = M_SQRT(err_sq / R(5.0));
        if (err_norm > R(1.0)) {
            real factor = rmax(R(0.2), R(0.9) * M_POW(err_norm, R(-0.2)));
            h *= factor;
            if (h < h_min) return 0;
            continue;
        }
        real r_prev = state[0], r_next = next_state[0];
        if (r_prev > r_capture && r_next <= r_capture) {
            real denom = r_next - r_prev;
            real frac = (denom == R(0.0)) ? R(1.0) : (r_capture - r_prev) / denom;
            frac = clip_scalar(frac, R(0.0), R(1.0));
            for (int i = 0; i < 5; ++i) state[i] = state[i] + frac * (next_state[i] - state[i]);
            lam += frac * h;
            event_status = -1;
            break;
        }
        if (r_prev < r_escape && r_next >= r_escape) {
            real denom = r_next - r_prev;
            real frac = (denom == R(0.0)) ? R(1.0) : (r_escape - r_prev) / denom;
            frac = clip_scalar(frac, R(0.0), R(1.0));
            for (int i = 0; i < 5; ++i) state[i] = state[i] + frac * (next_state[i] - state[i]);
            lam += frac * h;
            event_status = 1;
            break;
        }
        for (int i = 0; i < 5; ++i) state[i] = next_state[i];
        for (int i = 0; i < 5; ++i) k1[i] = k7[i];
        lam += h;
        if (!all_finite5(state)) return 0;
        if (err_norm < R(1e-10)) h *= R(5.0);
        else h *= rmin(R(5.0), R(0.9) * M_POW(err_norm, R(-0.2)));
    }
    return kerr_extract_angle(state, p_t, p_phi, M, a, r_capture, event_status, fa_out, nh_out);
}

/* metrics.py:570-658, fixed-step (radius-banded) RK4 -- the GPU fp32 kernel's spec */
static int kerr_trace_rk4(real M, real a, real r_plus, real r_obs, real alpha, real theta,
                          real theta_obs, real lambda_max, real h_max, int axis_refine,
                          real *fa_out, int64_t *nh_out)
{
    real state[5], p_t, p_phi;
    *fa_out = LTO_NAN; *nh_out = 0;
    if (!kerr_ic(M, a, r_obs, alpha, theta, theta_obs, state, &p_t, &p_phi)) return 0;
    real r_capture = r_plus * R(1.01);
    real r_escape = r_obs * R(2.0);
    real k1[5], k2[5], k3[5], k4[5], tmp[5], next_state[5];
    real lam = R(0.0);
    int event_status = 2;
    real h_base = h_max;
    if (axis_refine) h_base = rmin(h_base, R(0.5));
    real h_floor = rmin(axis_refine ? R(0.01) : R(0.02), h_base);
    while (lam < lambda_max) {
        real h = h_base;
        real remaining = lambda_max - lam;
        if (remaining < h) h = remaining;
        if (h <= R(0.0)) break;
        real r_curr = state[0];
        if (r_curr < r_capture * R(4.0)) h = rmin(h, axis_refine ? R(0.20) : R(0.25));
        if (r_curr < r_capture * R(2.0)) h = rmin(h, axis_refine ? R(0.08) : R(0.10));
        if (r_curr < r_capture * R(1.2)) h = rmin(h, axis_refine ? R(0.03) : R(0.05));
        real r_prev = state[0];
        for (;;) {
            rk4_step_kerr(state, h, p_t, p_phi, M, a, r_plus, k1, k2, k3, k4, tmp, next_state);
            if (all_finite5(next_state) && next_state[0] > R(0.0)) break;
            if (h <= h_floor) return 0;
            h *= R(0.5);
        }
        real r_next = next_state[0];
        if (r_prev > r_capture && r_next <= r_capture) {
            real denom = r_next - r_prev;
            real frac = (denom == R(0.0)) ? R(1.0) : (r_capture - r_prev) / denom;
            frac = clip_scalar(frac, R(0.0), R(1.0));
            for (int i = 0; i < 5; ++i) state[i] = state[i] + frac * (next_state[i] - state[i]);
            lam += frac * h;
            event_status = -1;
            break;
        }
        if (r_prev < r_escape && r_next >= r_escape) {
            real denom = r_next - r_prev;
            real frac = (denom == R(0.0)) ? R(1.0) : (r_escape - r_prev) / denom;
            frac = clip_scalar(frac, R(0.0), R(1.0));
            for (int i = 0; i < 5; ++i) state[i] = state[i] + frac * (next_state[i] - state[i]);
            lam += frac * h;
            event_status = 1;
            break;
        }
        for (int i = 0; i < 5; ++i) state[i] = next_state[i];
        lam += h;
        if (!all_finite5(state)) return 0;
    }
    return kerr_extract_angle(state, p_t, p_phi, M, a, r_capture, event_status, fa_out, nh_out);
}

/* ------------------------------------------------------------------------ */
/* exported: single-function probes (fixtures F1, F2)                        */
/* ------------------------------------------------------------------------ */
void LTO_NAME(lto_kerr_rhs)(const real *state5, real p_t, real p_phi, real M, real a, real r_plus,
                            real *out5)
{
    kerr_rhs(state5, p_t, p_phi, M, a, r_plus, out5);
}

int LTO_NAME(lto_kerr_ic)(real M, real a, real r_obs, real alpha, real theta, real theta_obs,
                          real *state5, real *p_t, real *p_phi)
{
    return kerr_ic(M, a, r_obs, alpha, theta, theta_obs, state5, p_t, p_phi);
}

/* ------------------------------------------------------------------------ */
/* exported: batch drivers (twins of metrics.py:661-668 and :671-679)        */
/* integrator: 0 = DP45 (reference production), 1 = RK4 (metrics.py:570-658) */
/* out_status / out_evals may be NULL.                                       */
/* ------------------------------------------------------------------------ */
int LTO_NAME(lto_trace_batch_schw)(real M, real r_obs, const real *alphas, int64_t n,
                                   real phi_max, real h_max, real *out_fa, int64_t *out_w,
                                   int8_t *out_status, uint32_t *out_evals)
{
    real R_S = R(2.0) * M;
#pragma omp parallel for schedule(dynamic, 256)
    for (int64_t i = 0; i < n; ++i) {
        real fa; int64_t nh;
        lto_evals = 0;
        int s = schw_trace_ray(M, R_S, r_obs, alphas[i], phi_max, h_max, &fa, &nh);
        out_fa[i] = (s == 1) ? fa : LTO_NAN;
        out_w[i] = nh;
        if (out_status) out_status[i] = (int8_t)s;
        if (out_evals) out_evals[i] = lto_evals;
    }
    return 0;
}

int LTO_NAME(lto_trace_batch_kerr)(real M, real a, real r_obs, const real *alphas,
                                   const real *thetas, real theta_obs, real lambda_max,
                                   const uint8_t *axis_refines, int integrator, int64_t n,
                                   real *out_fa, int64_t *out_w, int8_t *out_status,
                                   uint32_t *out_evals)
{
    if (M_FABS(a) > M) return -1;
    real r_plus = M + M_SQRT(M * M - a * a);
#pragma omp parallel for schedule(dynamic, 64)
    for (int64_t i = 0; i < n; ++i) {
        real fa; int64_t nh;
        lto_evals = 0;
        int ar = axis_refines ? (axis_refines[i] != 0) : 0;
        int s = integrator == 0
            ? kerr_trace_dp45(M, a, r_plus, r_obs, alphas[i], thetas[i], theta_obs, lambda_max,
                              R(1.0), ar, &fa, &nh)
            : kerr_trace_rk4(M, a, r_plus, r_obs, alphas[i], thetas[i], theta_obs, lambda_max,
                             R(1.0), ar, &fa, &nh);
        out_fa[i] = (s == 1) ? fa : LTO_NAN;
        out_w[i] = nh;
        if (out_status) out_status[i] = (int8_t)s;
        if (out_evals) out_evals[i] = lto_evals;
    }
    return 0;
}

#ifndef LTO_F32
/* ------------------------------------------------------------------------ */
/* camera / lookup / colouring (float64 only)                                */
/* ------------------------------------------------------------------------ */

/* image_lens.py:21-61: d, e_x, e_y, in_front from psi = (pitch_up, yaw_right) */
void lto_psi_frame(double psi_y, double psi_x, double *d, double *e_x, double *e_y, int *in_front)
{
    double sin_pitch = sin(psi_y), cos_pitch = cos(psi_y);
    double sin_yaw = sin(psi_x), cos_yaw = cos(psi_x);
    d[0] = sin_yaw * cos_pitch; d[1] = -sin_pitch; d[2] = cos_yaw * cos_pitch;
    *in_front = d[2] > 1e-12;
    const double cam_x[3] = {1.0, 0.0, 0.0}, cam_y[3] = {0.0, 1.0, 0.0};
    double dot = cam_x[0] * d[0] + cam_x[1] * d[1] + cam_x[2] * d[2];
    for (int i = 0; i < 3; ++i) e_x[i] = cam_x[i] - dot * d[i];
    double nrm = sqrt(e_x[0] * e_x[0] + e_x[1] * e_x[1] + e_x[2] * e_x[2]);
    if (nrm < 1e-12) {
        dot = cam_y[0] * d[0] + cam_y[1] * d[1] + cam_y[2] * d[2];
        for (int i = 0; i < 3; ++i) e_x[i] = cam_y[i] - dot * d[i];
        nrm = sqrt(e_x[0] * e_x[0] + e_x[1] * e_x[1] + e_x[2] * e_x[2]);
    }
    double dn = nrm > 1e-12 ? nrm : 1e-12;
    for (int i = 0; i < 3; ++i) e_x[i] /= dn;
    double dyd = cam_y[0] * d[0] + cam_y[1] * d[1] + cam_y[2] * d[2];
    double dye = cam_y[0] * e_x[0] + cam_y[1] * e_x[1] + cam_y[2] * e_x[2];
    for (int i = 0; i < 3; ++i) e_y[i] = cam_y[i] - dyd * d[i] - dye * e_x[i];
    nrm = sqrt(e_y[0] * e_y[0] + e_y[1] * e_y[1] + e_y[2] * e_y[2]);
    if (nrm < 1e-12) {
        e_y[0] = d[1] * e_x[2] - d[2] * e_x[1];
        e_y[1] = d[2] * e_x[0] - d[0] * e_x[2];
        e_y[2] = d[0] * e_x[1] - d[1] * e_x[0];
        nrm = sqrt(e_y[0] * e_y[0] + e_y[1] * e_y[1] + e_y[2] * e_y[2]);
    }
    dn = nrm > 1e-12 ? nrm : 1e-12;
    for (int i = 0; i < 3; ++i) e_y[i] /= dn;
}

/* image_lens.py:133-152 (alpha, stored float32: quirk Q2), :193-208 (theta, float64),
 * :210-216 (axis-refine columns).  Pixel corners, no +0.5. */
void lto_pixel_angles(int H, int W, double hfov, double vfov, double psi_y, double psi_x,
                      double axis_refine_frac, float *alpha_out, double *theta_out,
                      uint8_t *axis_cols_out)
{
    double d[3], e_x[3], e_y[3]; int front;
    lto_psi_frame(psi_y, psi_x, d, e_x, e_y, &front);
    double fx = (W / 2.0) / tan(hfov / 2);
    double fy = (H / 2.0) / tan(vfov / 2);
#pragma omp parallel for schedule(static)
    for (int iy = 0; iy < H; ++iy) {
        double y_cam = (iy - H / 2.0) / fy;
        for (int ix = 0; ix < W; ++ix) {
            double x_cam = (ix - W / 2.0) / fx;
            double denom = sqrt(1.0 + x_cam * x_cam + y_cam * y_cam);
            double cos_alpha = ((x_cam * d[0]) + (y_cam * d[1]) + d[2]) / denom;
            if (cos_alpha < -1.0) cos_alpha = -1.0;
            if (cos_alpha > 1.0) cos_alpha = 1.0;
            if (alpha_out) alpha_out[(size_t)iy * W + ix] = (float)acos(cos_alpha);
            if (theta_out) {
                double vx = x_cam / denom, vy = y_cam / denom, vz = 1.0 / denom;
                theta_out[(size_t)iy * W + ix] =
                    atan2(vx * e_x[0] + vy * e_x[1] + vz * e_x[2],
                          vx * e_y[0] + vy * e_y[1] + vz * e_y[2]);
            }
        }
    }
    if (axis_cols_out) {
        if (front) {
            double bh_x = d[0] / d[2];
            double m = 0.0;
            for (int ix = 0; ix < W; ++ix) {
                double xr = fabs((ix - W / 2.0) / fx - bh_x);
                if (xr > m) m = xr;
            }
            if (m < 1e-12) m = 1e-12;
            for (int ix = 0; ix < W; ++ix)
                axis_cols_out[ix] = fabs((ix - W / 2.0) / fx - bh_x) <= axis_refine_frac * m;
        } else {
            memset(axis_cols_out, 0, (size_t)W);
        }
    }
}

/* image_lens.py:155-178 (spherically symmetric) and :185-280 (2-D) in one:
 *   kind 0 Schwarzschild (every pixel traced, phi_max 50, h 0.05),
 *   kind 1 Kerr (integrator 0 DP45 / 1 RK4, lambda_max = max(5000, 6 r_obs)).
 * tb_symmetry: 0 off (every row traced), 1 reference behaviour (quirk Q1: rows
 * 0..(H+1)/2-1 traced, copied to rows H-1-j), applied only when the reference
 * would apply it (Kerr, theta_obs ~ pi/2, psi_y ~ 0).
 * Outputs: fa float32 (NaN unless escaped), winding uint16 (clipped), status int8
 * (nullable), evals uint32 (nullable).  Returns the number of rays traced. */
int64_t lto_lookup(int kind, double M, double a, double r_obs, double theta_obs,
                   int H, int W, double hfov, double vfov, double psi_y, double psi_x,
                   double axis_refine_frac, int integrator, int tb_symmetry,
                   float *out_fa, uint16_t *out_w, int8_t *out_status, uint32_t *out_evals)
{
    size_t n = (size_t)H * W;
    float *alpha32 = (float *)malloc(n * sizeof(float));
    double *theta = (double *)malloc(n * sizeof(double));
    uint8_t *cols = (uint8_t *)malloc((size_t)W);
    lto_pixel_angles(H, W, hfov, vfov, psi_y, psi_x, axis_refine_frac, alpha32, theta, cols);
    int use_tb = 0;
    if (kind == 1 && tb_symmetry) {
        /* np.isclose(x, y): |x-y| <= 1e-8 + 1e-5*|y| */
        use_tb = fabs(theta_obs - M_PI / 2) <= 1e-8 + 1e-5 * (M_PI / 2) && fabs(psi_y) <= 1e-8;
    }
    int trace_rows = use_tb ? (H + 1) / 2 : H;
    size_t nt = (size_t)trace_rows * W;
    double *al = (double *)malloc(nt * sizeof(double));
    uint8_t *ar = (uint8_t *)malloc(nt);
    double *fa = (double *)malloc(nt * sizeof(double));
    int64_t *w = (int64_t *)malloc(nt * sizeof(int64_t));
    int8_t *st = (int8_t *)malloc(nt);
    uint32_t *ev = (uint32_t *)malloc(nt * sizeof(uint32_t));
    for (size_t i = 0; i < nt; ++i) { al[i] = (double)alpha32[i]; ar[i] = cols[i % W]; }
    if (kind == 0)
        lto_trace_batch_schw(M, r_obs, al, (int64_t)nt, 50.0, 0.05, fa, w, st, ev);
    else {
        double lam = 6.0 * r_obs > 5000.0 ? 6.0 * r_obs : 5000.0;
        lto_trace_batch_kerr(M, a, r_obs, al, theta, theta_obs, lam, ar, integrator, (int64_t)nt,
                             fa, w, st, ev);
    }
    for (size_t i = 0; i < n; ++i) {
        out_fa[i] = NAN; out_w[i] = 0;
        if (out_status) out_status[i] = 0;
        if (out_evals) out_evals[i] = 0;
    }
    for (size_t i = 0; i < nt; ++i) {
        out_fa[i] = (float)fa[i];
        int64_t ww = w[i] < 0 ? 0 : (w[i] > 65535 ? 65535 : w[i]);
        out_w[i] = (uint16_t)ww;
        if (out_status) out_status[i] = st[i];
        if (out_evals) out_evals[i] = ev[i];
    }
    if (use_tb) {
        int top_half = H / 2;
        for (int j = 0; j < top_half; ++j) {
            size_t dst = (size_t)(H - 1 - j) * W, src = (size_t)j * W;
            memcpy(out_fa + dst, out_fa + src, (size_t)W * sizeof(float));
            memcpy(out_w + dst, out_w + src, (size_t)W * sizeof(uint16_t));
            if (out_status) memcpy(out_status + dst, out_status + src, (size_t)W);
            if (out_evals) memcpy(out_evals + dst, out_evals + src, (size_t)W * sizeof(uint32_t));
        }
    }
    free(alpha32); free(theta); free(cols); free(al); free(ar); free(fa); free(w); free(st); free(ev);
    return (int64_t)nt;
}

static const float WINDING_COLORS[5][3] = { /* image_lens.py:287-293 */
    {0.0f, 0.2f, 1.0f}, {0.0f, 0.7f, 1.0f}, {0.0f, 1.0f, 0.4f}, {1.0f, 1.0f, 0.0f}, {1.0f, 0.4f, 0.0f}};

/* image_lens.py:296-397.  source/out: (H, W, C) float32, C = 1 (grayscale) or 3.
 * np.rint = round-half-even = rint() in the default rounding mode. */
void lto_render(const float *source, int H, int W, int C, const float *fa_lookup,
                const uint16_t *winding, double hfov, double vfov, double psi_y, double psi_x,
                int loop_around, float *out)
{
    double d[3], e_x[3], e_y[3]; int front_bh;
    lto_psi_frame(psi_y, psi_x, d, e_x, e_y, &front_bh);
    double fx = (W / 2.0) / tan(hfov / 2);
    double fy = (H / 2.0) / tan(vfov / 2);
    float luma_colors[5];
    for (int k = 0; k < 5; ++k) /* float32 matmul WINDING_COLORS @ luma */
        luma_colors[k] = WINDING_COLORS[k][0] * 0.299f + WINDING_COLORS[k][1] * 0.587f
                       + WINDING_COLORS[k][2] * 0.114f;
#pragma omp parallel for schedule(static)
    for (int iy = 0; iy < H; ++iy) {
        double y_cam = (iy - H / 2.0) / fy;
        for (int ix = 0; ix < W; ++ix) {
            size_t p = (size_t)iy * W + ix;
            float *o = out + p * C;
            for (int c = 0; c < C; ++c) o[c] = 0.0f;
            float faf = fa_lookup[p];
            if (!isfinite(faf)) continue;
            /* NEP-50 weak scalar: final_alpha_lookup (f32) > np.pi/2 compares in float32 */
            float half_pi_f = (float)(M_PI / 2);
            if (faf > half_pi_f) {
                int idx = winding ? winding[p] : 0;
                if (idx > 4) idx = 4;
                if (C == 1) o[0] = luma_colors[idx];
                else for (int c = 0; c < 3 && c < C; ++c) o[c] = WINDING_COLORS[idx][c];
                continue;
            }
            double x_cam = (ix - W / 2.0) / fx;
            double denom = sqrt(1.0 + x_cam * x_cam + y_cam * y_cam);
            double vx = x_cam / denom, vy = y_cam / denom, vz = 1.0 / denom;
            double th = atan2(vx * e_x[0] + vy * e_x[1] + vz * e_x[2],
                              vx * e_y[0] + vy * e_y[1] + vz * e_y[2]);
            double fa = (double)faf;
            double sin_fa = sin(fa), cos_fa = cos(fa), sin_th = sin(th), cos_th = cos(th);
            double src_vx = cos_fa * d[0] + sin_fa * (sin_th * e_x[0] + cos_th * e_y[0]);
            double src_vy = cos_fa * d[1] + sin_fa * (sin_th * e_x[1] + cos_th * e_y[1]);
            double src_vz = cos_fa * d[2] + sin_fa * (sin_th * e_x[2] + cos_th * e_y[2]);
            int front = src_vz > 1e-12;
            if (loop_around) {
                double sxc = 0.0, syc = 0.0;
                if (front) { sxc = src_vx / src_vz; syc = src_vy / src_vz; }
                long long sx = (long long)rint(sxc * fx + W / 2.0);
                long long sy = (long long)rint(syc * fy + H / 2.0);
                sx %= W; if (sx < 0) sx += W;   /* numpy % is floor-mod */
                sy %= H; if (sy < 0) sy += H;
                for (int c = 0; c < C; ++c) o[c] = source[((size_t)sy * W + sx) * C + c];
            } else {
                long long sx = -1, sy = -1;
                if (front) {
                    sx = (long long)rint(src_vx / src_vz * fx + W / 2.0);
                    sy = (long long)rint(src_vy / src_vz * fy + H / 2.0);
                }
                if (front && sy >= 0 && sy < H && sx >= 0 && sx < W) {
                    for (int c = 0; c < C; ++c) o[c] = source[((size_t)sy * W + sx) * C + c];
                } else { /* magenta */
                    if (C == 1) o[0] = 1.0f;
                    else { o[0] = 1.0f; if (C > 2) o[2] = 1.0f; }
                }
            }
        }
    }
}

/* matplotlib.image.imsave path used at image_lens.py:510: float RGB in [0,1] ->
 * uint8 by (x * 255).astype(uint8) in float32 (truncation), alpha = 255. */
void lto_rgba8(const float *rgb, int64_t n_pix, int C, uint8_t *rgba)
{
#pragma omp parallel for schedule(static)
    for (int64_t p = 0; p < n_pix; ++p) {
        for (int c = 0; c < 3; ++c) {
            float v = rgb[p * C + (C == 1 ? 0 : c)] * 255.0f;
            rgba[p * 4 + c] = (uint8_t)v;
        }
        rgba[p * 4 + 3] = 255;
    }
}

/* black_hole_shadow.py:7-15, :32-37 -- analytic 0/1 shadow, image[i, j] with i = x */
void lto_shadow_analytic(int width, int height, double fov, double alpha_crit, double *image)
{
    for (int j = 0; j < height; ++j)
        for (int i = 0; i < width; ++i) {
            double iu = (i - width / 2.0) / (width / 2.0);
            double ju = (j - height / 2.0) / (height / 2.0);
            double ax = atan(iu * tan(fov / 2)), ay = atan(ju * tan(fov / 2));
            double alpha = acos(cos(ax) * cos(ay));
            image[(size_t)i * height + j] = alpha < alpha_crit ? 0.0 : 1.0;
        }
}

int lto_num_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* Number of OpenMP threads the batch loops use from now on (bench.py's cpu_baseline: one per usable CPU of the
 * job, which on a shared host is fewer than the CPUs the process can see). */
void lto_set_num_threads(int n)
{
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}
#endif /* !LTO_F32 */
