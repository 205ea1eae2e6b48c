/*
 * oracle/lt_oracle_dense.c  --  TEST INFRASTRUCTURE, NOT PRODUCT CODE (see lt_oracle.c).
 *
 * CPU restatement of the reference's dense single-ray path: geodesic_tracer.integrate_geodesic
 * (geodesic_tracer.py:22-71), i.e. the 8-D Hamiltonian right-hand sides of metrics.py driven by
 * scipy.integrate.solve_ivp(method='RK45', max_step=1, rtol=1e-8, atol=1e-10, two terminal radius
 * events, dense_output=True).
 *
 * The integrator lives in a third-party dependency that is not part of the reference tree:
 * scipy (requirements.txt:2, unpinned; 1.15.3 in the build container).  What is restated here is its
 * published algorithm: the Dormand-Prince 5(4) pair with Shampine's 4th-order dense output, Hairer's
 * initial step (Hairer, Norsett, Wanner, "Solving ODEs I", II.4), the step controller
 * h <- h * min(10, max(0.2, 0.9 err^(-1/5))) on an RMS error norm scaled by atol + rtol*max(|y|,|y_new|),
 * and events located with Brent's method on the dense output to 4 eps.
 *
 * Parity status: PINNED by tests/golden/dense_tracks.npz (F10): 59 tracks produced by the reference's
 * own integrate_geodesic in the build container -- every accepted point, the event point, nfev and
 * the outcome -- and 192 evaluations of the two 8-D right-hand sides on random states.
 */
#define _GNU_SOURCE
#include <float.h>
#include <math.h>
#include <stdint.h>
#include <string.h>

/* metrics.py:763-790  Schwarzschild.geodesic_equations (M carried as R_S = 2M) */
static void rhs8_schw(double M, const double *s, double *o)
{
    const double R_S = 2.0 * M;
    double r = s[1], th = s[2], p_t = s[4], p_r = s[5], p_th = s[6], p_phi = s[7];
    if (r <= R_S * 1.001) { memset(o, 0, 8 * sizeof(double)); return; }
    double f = 1.0 - R_S / r;
    double sin_th = sin(th), cos_th = cos(th);
    double sin_th_sq = sin_th * sin_th;
    if (sin_th_sq < 1e-15) sin_th_sq = 1e-15;
    double r2 = r * r;
    o[0] = -p_t / f;
    o[1] = f * p_r;
    o[2] = p_th / r2;
    o[3] = p_phi / (r2 * sin_th_sq);
    o[4] = 0.0;
    o[5] = (-(R_S / (2 * r2)) * (p_t * p_t / (f * f)) - (R_S / (2 * r2)) * (p_r * p_r)
            + (p_th * p_th + p_phi * p_phi / sin_th_sq) / pow(r, 3.0));
    o[6] = cos_th * (p_phi * p_phi) / (r2 * sin_th_sq * sin_th);
    o[7] = 0.0;
}

/* metrics.py:946-1029  Kerr.geodesic_equations */
static void rhs8_kerr(double M, double a, const double *s, double *o)
{
    double r = s[1], th = s[2], p_t = s[4], p_r = s[5], p_th = s[6], p_phi = s[7];
    double r_plus = M + sqrt(M * M - a * a);
    if (r <= r_plus * 1.001) { memset(o, 0, 8 * sizeof(double)); return; }
    double sin_th = sin(th), cos_th = cos(th);
    double r2 = r * r, a2 = a * a, s2 = sin_th * sin_th;
    double Sigma = r2 + a2 * (cos_th * cos_th);
    double Delta = r2 - 2 * M * r + a2;
    double A = (r2 + a2) * (r2 + a2) - a2 * Delta * s2;
    double SD = Sigma * Delta;
    double g_tt = -A / SD, g_tphi = -2 * M * a * r / SD, g_rr = Delta / Sigma, g_thth = 1.0 / Sigma;
    double g_pp = (Delta - a2 * s2) / (SD * s2);
    o[0] = g_tt * p_t + g_tphi * p_phi;
    o[1] = g_rr * p_r;
    o[2] = g_thth * p_th;
    o[3] = g_tphi * p_t + g_pp * p_phi;
    double dS_dr = 2 * r, dD_dr = 2 * r - 2 * M;
    double dA_dr = 4 * r * (r2 + a2) - a2 * dD_dr * s2;
    double SD2 = SD * SD, S2 = Sigma * Sigma;
    double mix = dS_dr * Delta + Sigma * dD_dr;
    double dg_tt_dr = -(dA_dr * Sigma * Delta - A * mix) / SD2;
    double dg_tphi_dr = -(2 * M * a * (SD - r * mix)) / SD2;
    double dg_rr_dr = (dD_dr * Sigma - Delta * dS_dr) / S2;
    double dg_thth_dr = -dS_dr / S2;
    double den = SD * s2;
    double dg_pp_dr = (dD_dr * Sigma * Delta * s2 - (Delta - a2 * s2) * mix * s2) / (den * den);
    o[5] = -0.5 * (dg_tt_dr * (p_t * p_t) + 2 * dg_tphi_dr * p_t * p_phi + dg_rr_dr * (p_r * p_r)
                   + dg_thth_dr * (p_th * p_th) + dg_pp_dr * (p_phi * p_phi));
    double dS_dth = -2 * a2 * sin_th * cos_th;
    double dA_dth = -a2 * Delta * 2 * sin_th * cos_th;
    double dg_tt_dth = -(dA_dth * Sigma * Delta - A * dS_dth * Delta) / SD2;
    double dg_tphi_dth = 2 * M * a * r * dS_dth / (S2 * Delta);
    double dg_rr_dth = -Delta * dS_dth / S2;
    double dg_thth_dth = -dS_dth / S2;
    double num = Delta - a2 * s2;
    double dnum = -a2 * 2 * sin_th * cos_th;
    double dden = dS_dth * Delta * s2 + Sigma * Delta * 2 * sin_th * cos_th;
    double dg_pp_dth = (dnum * den - num * dden) / (den * den);
    o[6] = -0.5 * (dg_tt_dth * (p_t * p_t) + 2 * dg_tphi_dth * p_t * p_phi + dg_rr_dth * (p_r * p_r)
                   + dg_thth_dth * (p_th * p_th) + dg_pp_dth * (p_phi * p_phi));
    o[4] = 0.0;
    o[7] = 0.0;
}

/* kind 0 = Schwarzschild class, 1 = Kerr class (the reference keeps two formulas; so does the oracle) */
void lto_rhs8(int kind, double M, double a, const double *state, double *out)
{
    if (kind == 0) rhs8_schw(M, state, out);
    else rhs8_kerr(M, a, state, out);
}

/* ------------------------------------------------------------------------------------------------
 * scipy RK45 (Dormand-Prince) restated
 * ---------------------------------------------------------------------------------------------- */
/* the system is autonomous: the stage abscissae c_i are not needed */
static const double DP_A[6][5] = {{0, 0, 0, 0, 0},
                                  {1.0 / 5, 0, 0, 0, 0},
                                  {3.0 / 40, 9.0 / 40, 0, 0, 0},
                                  {44.0 / 45, -56.0 / 15, 32.0 / 9, 0, 0},
                                  {19372.0 / 6561, -25360.0 / 2187, 64448.0 / 6561, -212.0 / 729, 0},
                                  {9017.0 / 3168, -355.0 / 33, 46732.0 / 5247, 49.0 / 176, -5103.0 / 18656}};
static const double DP_B[6] = {35.0 / 384, 0, 500.0 / 1113, 125.0 / 192, -2187.0 / 6784, 11.0 / 84};
static const double DP_E[7] = {-71.0 / 57600, 0, 71.0 / 16695, -71.0 / 1920, 17253.0 / 339200, -22.0 / 525, 1.0 / 40};
static const double DP_P[7][4] = {
    {1, -8048581381.0 / 2820520608, 8663915743.0 / 2820520608, -12715105075.0 / 11282082432},
    {0, 0, 0, 0},
    {0, 131558114200.0 / 32700410799, -68118460800.0 / 10900136933, 87487479700.0 / 32700410799},
    {0, -1754552775.0 / 470086768, 14199869525.0 / 1410260304, -10690763975.0 / 1880347072},
    {0, 127303824393.0 / 49829197408, -318862633887.0 / 49829197408, 701980252875.0 / 199316789632},
    {0, -282668133.0 / 205662961, 2019193451.0 / 616988883, -1453857185.0 / 822651844},
    {0, 40617522.0 / 29380423, -110615467.0 / 29380423, 69997945.0 / 29380423}};

typedef struct { int kind; double M, a; int64_t nfev; } Sys;

static void fun(Sys *S, const double *y, double *o)
{
    ++S->nfev;
    lto_rhs8(S->kind, S->M, S->a, y, o);
}

static double rms8(const double *x)
{
    double s = 0;
    for (int i = 0; i < 8; ++i) s += x[i] * x[i];
    return sqrt(s) / sqrt(8.0);
}

/* dense output of the last step: y(t) = y_old + h * Q [x, x^2, x^3, x^4], x = (t - t_old) / h */
static void dense_eval(const double *y_old, double t_old, double h, double Q[8][4], double t, double *y)
{
    double x = (t - t_old) / h, p[4];
    p[0] = x; p[1] = p[0] * x; p[2] = p[1] * x; p[3] = p[2] * x;
    for (int i = 0; i < 8; ++i) {
        double s = 0;
        for (int j = 0; j < 4; ++j) s += Q[i][j] * p[j];
        y[i] = y_old[i] + h * s;
    }
}

typedef struct { const double *y_old; double t_old, h; double (*Q)[4]; double radius; } EvCtx;

static double ev_fun(const EvCtx *c, double t)
{
    double y[8];
    dense_eval(c->y_old, c->t_old, c->h, c->Q, t, y);
    return y[1] - c->radius;
}

/* Brent's method (Brent 1973, ch. 4) with xtol = rtol = 4 eps, as solve_ivp asks of brentq */
static double brent_root(const EvCtx *c, double xa, double xb)
{
    const double xtol = 4 * DBL_EPSILON, rtol = 4 * DBL_EPSILON;
    double xpre = xa, xcur = xb, xblk = 0, fpre = ev_fun(c, xpre), fcur = ev_fun(c, xcur), fblk = 0, spre = 0, scur = 0;
    if (fpre == 0) return xpre;
    if (fcur == 0) return xcur;
    for (int it = 0; it < 100; ++it) {
        if (fpre != 0 && fcur != 0 && (signbit(fpre) != signbit(fcur))) {
            xblk = xpre; fblk = fpre;
            spre = scur = xcur - xpre;
        }
        if (fabs(fblk) < fabs(fcur)) {
            xpre = xcur; xcur = xblk; xblk = xpre;
            fpre = fcur; fcur = fblk; fblk = fpre;
        }
        double delta = (xtol + rtol * fabs(xcur)) / 2;
        double sbis = (xblk - xcur) / 2;
        if (fcur == 0 || fabs(sbis) < delta) return xcur;
        if (fabs(spre) > delta && fabs(fcur) < fabs(fpre)) {
            double stry;
            if (xpre == xblk) {
                stry = -fcur * (xcur - xpre) / (fcur - fpre);
            } else {
                double dpre = (fpre - fcur) / (xpre - xcur), dblk = (fblk - fcur) / (xblk - xcur);
                stry = -fcur * (fblk * dblk - fpre * dpre) / (dblk * dpre * (fblk - fpre));
            }
            if (2 * fabs(stry) < fmin(fabs(spre), 3 * fabs(sbis) - delta)) { spre = scur; scur = stry; }
            else { spre = sbis; scur = sbis; }
        } else {
            spre = sbis; scur = sbis;
        }
        xpre = xcur; fpre = fcur;
        if (fabs(scur) > delta) xcur += scur;
        else xcur += (sbis > 0 ? delta : -delta);
        fcur = ev_fun(c, xcur);
    }
    return xcur;
}

/*
 * One track.  out_t[max_points], out_y[8][max_points] row-major (component-major, like solution.y).
 * Returns the number of points a complete record holds (may exceed max_points: then only the first
 * max_points are stored, the LAST stored slot always holding the final point).
 * status: 1 capture event, 2 escape event, 0 lambda_max reached, -1 step size underflow.
 */
/* Optional per-step log (accepted steps only): step length and the error norm it was accepted with.  Used by
 * lto_predict_attempts below -- the CPU twin of the dense kernel's length predictor -- and never by the record path. */
static __thread double *g_step_h = NULL, *g_step_err = NULL;
static __thread int64_t g_step_cap = 0, g_step_n = 0;

int64_t lto_integrate_dense(int kind, double M, double a, const double *state0, double lambda_max, double r_stop_inner,
                            double r_stop_outer, double rtol, double atol, double max_step, int64_t max_points,
                            double *out_t, double *out_y, int *out_status, int64_t *out_nfev)
{
    Sys S = {kind, M, a, 0};
    double t = 0, y[8], f[8], K[7][8], Q[8][4];
    memcpy(y, state0, sizeof y);
    int64_t n = 0;
#define PUSH(tt, yy)                                                                                                  \
    do {                                                                                                              \
        int64_t slot = n < max_points ? n : max_points - 1;                                                           \
        if (max_points > 0) { out_t[slot] = (tt); for (int c_ = 0; c_ < 8; ++c_) out_y[c_ * max_points + slot] = (yy)[c_]; } \
        ++n;                                                                                                          \
    } while (0)
    PUSH(t, y);
    fun(&S, y, f);
    /* select_initial_step (Hairer II.4), order of the error estimator = 4 */
    double h_abs;
    {
        double sc[8], v[8], y1[8], f1[8];
        for (int i = 0; i < 8; ++i) sc[i] = atol + fabs(y[i]) * rtol;
        for (int i = 0; i < 8; ++i) v[i] = y[i] / sc[i];
        double d0 = rms8(v);
        for (int i = 0; i < 8; ++i) v[i] = f[i] / sc[i];
        double d1 = rms8(v);
        double h0 = (d0 < 1e-5 || d1 < 1e-5) ? 1e-6 : 0.01 * d0 / d1;
        h0 = fmin(h0, lambda_max);
        for (int i = 0; i < 8; ++i) y1[i] = y[i] + h0 * f[i];
        fun(&S, y1, f1);
        for (int i = 0; i < 8; ++i) v[i] = (f1[i] - f[i]) / sc[i];
        double d2 = rms8(v) / h0;
        double h1 = (d1 <= 1e-15 && d2 <= 1e-15) ? fmax(1e-6, h0 * 1e-3) : pow(0.01 / fmax(d1, d2), 1.0 / 5);
        h_abs = fmin(fmin(100 * h0, h1), fmin(lambda_max, max_step));
    }
    double g_in = y[1] - r_stop_inner, g_out = y[1] - r_stop_outer;
    int status = 0;
    while (t != lambda_max) {
        double min_step = 10 * fabs(nextafter(t, INFINITY) - t);
        if (h_abs > max_step) h_abs = max_step;
        else if (h_abs < min_step) h_abs = min_step;
        int rejected = 0;
        double h, t_new, y_new[8], f_new[8];
        for (;;) {
            if (h_abs < min_step) { status = -1; goto done; }
            h = h_abs;
            t_new = t + h;
            if (t_new - lambda_max > 0) t_new = lambda_max;
            h = t_new - t;
            h_abs = fabs(h);
            memcpy(K[0], f, sizeof f);
            for (int s = 1; s < 6; ++s) {
                double ys[8];
                for (int i = 0; i < 8; ++i) {
                    double d = 0;
                    for (int j = 0; j < s; ++j) d += K[j][i] * DP_A[s][j];
                    ys[i] = y[i] + d * h;
                }
                fun(&S, ys, K[s]);
            }
            for (int i = 0; i < 8; ++i) {
                double d = 0;
                for (int j = 0; j < 6; ++j) d += K[j][i] * DP_B[j];
                y_new[i] = y[i] + h * d;
            }
            fun(&S, y_new, f_new);
            memcpy(K[6], f_new, sizeof f_new);
            double e[8];
            for (int i = 0; i < 8; ++i) {
                double d = 0;
                for (int j = 0; j < 7; ++j) d += K[j][i] * DP_E[j];
                e[i] = d * h / (atol + fmax(fabs(y[i]), fabs(y_new[i])) * rtol);
            }
            double err = rms8(e);
            if (err < 1) {
                if (g_step_h && g_step_n < g_step_cap) { g_step_h[g_step_n] = h; g_step_err[g_step_n] = err; }
                ++g_step_n;
                double factor = (err == 0) ? 10 : fmin(10, 0.9 * pow(err, -0.2));
                if (rejected) factor = fmin(1, factor);
                h_abs *= factor;
                break;
            }
            h_abs *= fmax(0.2, 0.9 * pow(err, -0.2)); /* a NaN error norm lands here too (nan < 1 is false) */
            rejected = 1;
        }
        /* step accepted: dense output coefficients, events */
        for (int i = 0; i < 8; ++i)
            for (int j = 0; j < 4; ++j) {
                double d = 0;
                for (int s = 0; s < 7; ++s) d += K[s][i] * DP_P[s][j];
                Q[i][j] = d;
            }
        double gn_in = y_new[1] - r_stop_inner, gn_out = y_new[1] - r_stop_outer;
        int hit_in = g_in >= 0 && gn_in <= 0, hit_out = g_out <= 0 && gn_out >= 0;
        if (hit_in || hit_out) {
            EvCtx c = {y, t, h, Q, 0};
            double root_in = 0, root_out = 0;
            if (hit_in) { c.radius = r_stop_inner; root_in = brent_root(&c, t, t_new); }
            if (hit_out) { c.radius = r_stop_outer; root_out = brent_root(&c, t, t_new); }
            int take_in = hit_in && (!hit_out || root_in <= root_out);
            double te = take_in ? root_in : root_out, ye[8];
            dense_eval(y, t, h, Q, te, ye);
            PUSH(te, ye);
            status = take_in ? 1 : 2;
            goto done;
        }
        g_in = gn_in; g_out = gn_out;
        t = t_new;
        memcpy(y, y_new, sizeof y);
        memcpy(f, f_new, sizeof f);
        PUSH(t, y);
    }
done:
#undef PUSH
    *out_status = status;
    *out_nfev = S.nfev;
    return n;
}

/* One track through the integrator above at a LOOSE tolerance with a free step size, logging (h, err) of every
 * accepted step: the raw material of the length predictor (lt_dense.hpp, k_dense_predict).  Returns the number of
 * accepted steps; out_h / out_err hold the first `cap` of them. */
int64_t lto_dense_step_log(int kind, double M, double a, const double *state0, double lambda_max, double r_stop_inner,
                           double r_stop_outer, double rtol, double atol, double max_step, int64_t cap, double *out_h,
                           double *out_err, int *out_status, int64_t *out_nfev)
{
    double t2[2], y2[16];
    g_step_h = out_h; g_step_err = out_err; g_step_cap = cap; g_step_n = 0;
    lto_integrate_dense(kind, M, a, state0, lambda_max, r_stop_inner, r_stop_outer, rtol, atol, max_step, 2, t2, y2, out_status,
                        out_nfev);
    int64_t n = g_step_n;
    g_step_h = g_step_err = NULL; g_step_cap = 0; g_step_n = 0;
    /* the step that contains the terminal event is used up to the event only */
    if (n > 0 && n <= cap && *out_status > 0) {
        double before = 0;
        for (int64_t i = 0; i + 1 < n; ++i) before += out_h[i];
        out_h[n - 1] = t2[1] - before;
    }
    return n;
}
