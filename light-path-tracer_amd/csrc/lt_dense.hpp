// lt_dense.hpp -- batched dense trajectories: the other side of the metric plugin.
//
// Replaces, for a whole batch of 8-D initial states, geodesic_tracer.integrate_geodesic
// (geodesic_tracer.py:22-71): solve_ivp(RK45, max_step, rtol, atol, two terminal radius events,
// dense_output) on metric.geodesic_equations (metrics.py:763-790 Schwarzschild, :946-1029 Kerr).
// One work-item per track, float64 throughout, every accepted point written to HBM.
//
// The integrator is scipy's (a dependency of the reference, not part of its tree); what is built here
// is its published algorithm -- Dormand-Prince 5(4), Hairer's initial step, the 0.9 err^(-1/5) controller
// on an RMS norm, Shampine's 4th-order dense output, Brent's method for the event time -- so that the
// step sequence (point count, nfev) equals solve_ivp's and the points agree to ~1e-9.
//
// Record layout: track-major --  t[i * max_points + p],  y[(i * max_points + p) * 8 + c] -- the
// layout of the reference's solution.t / solution.y per track.  Every lane appends one aligned 64-byte segment
// (four 16-byte stores) and one double per accepted point to ITS OWN record, so lanes need not be at the same
// point index: a lane that repeats a rejected attempt falls a point behind its neighbours without breaking
// anybody's store pattern.  (Round 1 used a point-major layout that coalesced only while all 64 tracks of a
// wavefront advanced in lockstep -- which made every lane wait for every rejected attempt of any other lane.)
#pragma once
#include "lt_device.hpp"

namespace lt {

struct DenseConsts {
    double M, a, a2, r_zero; // r_zero = 1.001 r_plus: the right-hand side is zero inside (metrics.py:767, :950)
    int floor_sin2;          // Schwarzschild class floors sin^2 at 1e-15 (metrics.py:774-775); Kerr's 8-D form does not
    double lambda_max, r_in, r_out; // r_in < 0: never; r_out < 0: twice the track's start radius (geodesic_tracer.py:45-46)
    double rtol, atol, max_step;
    int64_t n, max_points;
    int32_t max_attempts; // guard: a track that needs more step attempts ends with status -2 (solve_ivp has no such limit)
};

// p_t (component 4) and p_phi (component 7) are cyclic: their derivatives are identically zero, so a stage sum leaves them
// unchanged and their error estimate is exactly 0 -- in solve_ivp's arithmetic too (y + h * 0, 0 / scale).  Written out
// per component the compiler cannot drop that work without fast-math (0 * h is not 0 for a non-finite h), which was 1/4 of
// the stage sums and error scales; the integrators below skip the two components by hand.  Bit-identical.
__device__ __forceinline__ constexpr bool dense_cyclic(int c) { return c == 4 || c == 7; }

// 8-D Hamilton equations of H = g^{mu nu} p_mu p_nu / 2 in Boyer-Lindquist coordinates, state
// (t, r, theta, phi, p_t, p_r, p_theta, p_phi).  Same separable form as the 5-D one of lt_device.hpp
// (2 Sigma H = Delta p_r^2 + p_theta^2 + D^2 / sin^2 - P^2 / Delta, D = L - a E sin^2, P = E (r^2 + a^2) - a L)
// with the energy E = -p_t a state variable, and the exact gradient: the 2H terms are kept, so the function
// equals the reference's for any state, on shell or not.
__device__ __forceinline__ void rhs8_sc(const DenseConsts &k, const double *y, double s, double c, double *d)
{
    const double r = y[1], E = -y[4], pr = y[5], pth = y[6], L = y[7];
    if (r <= k.r_zero) {
#pragma unroll
        for (int i = 0; i < 8; ++i) d[i] = 0.0;
        return;
    }
    double s2 = s * s;
    if (k.floor_sin2 && s2 < 1e-15) s2 = 1e-15;
    const double r2 = r * r, ra = r2 + k.a2;
    const double Sigma = r2 + k.a2 * c * c, Delta = ra - 2.0 * k.M * r;
    // one division for the three reciprocals (each float64 division is ~12 instructions around a quarter-rate v_rcp_f64)
    const double SD = Sigma * Delta, t = M<double>::rcp_pos(SD * s2); // (> 0 outside r_zero; sin^2 = 0 gives NaN as the division's inf * 0 did)
    const double iS = (Delta * s2) * t, iD = (Sigma * s2) * t, is2 = SD * t;
    const double P = E * ra - k.a * L, D = L - k.a * E * s2;
    const double PD = P * iD, Ds = D * is2;
    const double F = Delta * pr * pr + pth * pth + D * Ds - P * PD; // 2 Sigma H
    const double H2 = F * iS;                                       // 2 H
    d[0] = (k.a * D + ra * PD) * iS;
    d[1] = Delta * pr * iS;
    d[2] = pth * iS;
    d[3] = (Ds + k.a * PD) * iS;
    d[4] = 0.0;
    // -(F_r - 2H Sigma_r) / (2 Sigma) and -(F_theta - 2H Sigma_theta) / (2 Sigma) with the factor 1/2 folded in
    d[5] = -iS * ((r - k.M) * (pr * pr + PD * PD) - r * (2.0 * E * PD + H2));
    d[6] = iS * (s * c) * (Ds * (2.0 * k.a * E + Ds) - H2 * k.a2);
    d[7] = 0.0;
}

__device__ __forceinline__ void rhs8(const DenseConsts &k, const double *y, double *d)
{
    double s, c;
    M<double>::sincos(y[2], s, c);
    rhs8_sc(k, y, s, c, d);
}

// right-hand side at a stage state whose polar angle is th0 + (y[2] - th0) with (s0, c0) = sincos(th0): the
// stages of a step sit within a fraction of a radian of its base point, so sin / cos come from a rotation of
// the base pair (sincos_shift; full evaluation per lane beyond 0.25 rad) instead of a fresh argument reduction
__device__ __forceinline__ void rhs8_near(const DenseConsts &k, const double *y, double th0, double s0, double c0,
                                          double *d, double &s, double &c)
{
    sincos_shift(th0, s0, c0, y[2] - th0, s, c);
    rhs8_sc(k, y, s, c, d);
}

__device__ __forceinline__ double rms8(const double *x)
{
    double s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += x[i] * x[i];
    return __builtin_sqrt(s) * 0.35355339059327373; // / sqrt(8)
}

// radius of the dense output y(t) = y_old + h Q [x, x^2, x^3, x^4] minus the event radius
__device__ __forceinline__ double dense_r(double r_old, double t_old, double h, const double *Qr, double t, double radius)
{
    double x = (t - t_old) / h;
    double p = ((Qr[3] * x + Qr[2]) * x + Qr[1]) * x + Qr[0];
    return r_old + h * (p * x) - radius;
}

// Brent's method on [xa, xb] to 4 eps (solve_ivp's brentq call)
__device__ inline double brent_root(double r_old, double t_old, double h, const double *Qr, double radius, double xa, double xb)
{
    const double tol = 4.0 * 2.220446049250313e-16;
    double xpre = xa, xcur = xb, xblk = 0, fblk = 0, spre = 0, scur = 0;
    double fpre = dense_r(r_old, t_old, h, Qr, xpre, radius), fcur = dense_r(r_old, t_old, h, Qr, xcur, radius);
    if (fpre == 0) return xpre;
    if (fcur == 0) return xcur;
    for (int it = 0; it < 100; ++it) {
        if (fpre != 0 && fcur != 0 && ((fpre < 0) != (fcur < 0))) {
            xblk = xpre; fblk = fpre;
            spre = scur = xcur - xpre;
        }
        if (fabs(fblk) < fabs(fcur)) {
            xpre = xcur; xcur = xblk; xblk = xpre;
            fpre = fcur; fcur = fblk; fblk = fpre;
        }
        double delta = (tol + tol * fabs(xcur)) * 0.5;
        double sbis = (xblk - xcur) * 0.5;
        if (fcur == 0 || fabs(sbis) < delta) return xcur;
        if (fabs(spre) > delta && fabs(fcur) < fabs(fpre)) {
            double stry;
            if (xpre == xblk) {
                stry = -fcur * (xcur - xpre) / (fcur - fpre);
            } else {
                double dpre = (fpre - fcur) / (xpre - xcur), dblk = (fblk - fcur) / (xblk - xcur);
                stry = -fcur * (fblk * dblk - fpre * dpre) / (dblk * dpre * (fblk - fpre));
            }
            if (2.0 * fabs(stry) < fmin(fabs(spre), 3.0 * fabs(sbis) - delta)) { spre = scur; scur = stry; }
            else { spre = sbis; scur = sbis; }
        } else {
            spre = sbis; scur = sbis;
        }
        xpre = xcur; fpre = fcur;
        xcur += (fabs(scur) > delta) ? scur : (sbis > 0 ? delta : -delta);
        fcur = dense_r(r_old, t_old, h, Qr, xcur, radius);
    }
    return xcur;
}

struct DenseOut {
    double *t;        // (n, max_points)
    double *y;        // (n, max_points, 8)
    int32_t *count;   // points a complete record holds (> max_points: truncated, last slot = final point)
    int8_t *status;   // 1 capture event, 2 escape event, 0 lambda_max reached, -1 step size underflow, -2 attempt limit
    int32_t *nfev;    // right-hand-side evaluations, counted like solve_ivp's nfev
};

#ifndef LT_DENSE_WAVES
#define LT_DENSE_WAVES 2 // waves per SIMD the register allocation is held to (measured: DESIGN.md 10)
#endif

// One track per lane.  Loop structure: solve_ivp's step is "repeat the attempt until it is accepted, then look
// for events"; written like that for 64 tracks at once (round 1), every lane waits while any other lane repeats an
// attempt and while any other lane locates its terminal event (Brent's method: several attempts' worth of
// instructions, once per track, the 64 tracks of a wave one after the other) -- measured lane utilisation 0.47.
// Here one loop iteration is ONE attempt of every lane: a lane that rejects retries in the next iteration while
// its neighbours go on with their next step (track-major records make that free), and a lane whose accepted step
// contains the terminal event leaves the loop; events are located after the loop by all lanes of the wave at the
// same time.  Each lane performs exactly the operations it did before, in the same order: results are
// bit-identical to round 1's.  What remains is the spread of track LENGTHS inside a wave: utilisation 0.59 against a
// bound of 0.63 for one track per lane (tools/dense_lane_stats.py).  A persistent-wave variant that refills idle
// lanes from a track queue was built and measured: utilisation 0.73, 7 % fewer instructions, but
// 11 % MORE time -- its service code (event location, track start-up) runs at 8/64 lanes and the denser stream
// clocks lower -- so it is not the kernel that ships.
// Blocks of 64 tracks are handed out from a queue head to a grid that fills the chip once, like the tiles of
// k_kerr_direct (the XCDs run at different clocks; a workgroup per block leaves the fastest idle at the end);
// head == nullptr: one workgroup per block.
//
// `perm` (nullable): slot -> track.  The length-binned launch (k_dense_predict + k_dense_window_sort below) passes the
// tracks ordered by predicted length inside windows of a few thousand, so that the 64 tracks of a wavefront end within
// a few attempts of each other; records, counts and endings are written at the TRACK's index either way, and a track's
// arithmetic never depends on its neighbours, so the output is byte-identical to the launch in caller order.
//
// Record writes.  The kernel issues the same number of cycles with its record stores compiled out (GRBM_GUI_ACTIVE,
// SQ_BUSY_CYCLES and SQ_WAVE_CYCLES equal to 0.3 %, tools/scratch/dense_wait_counters.sh): the stores are hidden
// completely.  What they cost is CLOCK: with the lanes filled by the length-binned order the chip held 2.38 GHz without
// the stores and 1.96 GHz with them (27.1 against 32.9 ms for 4 M tracks).  Measured where the bytes went: WRITE_SIZE
// 86.8 GB per launch for 51.9 GB of records -- the 8-byte affine-parameter value of a point, stored alone, left L2 as a
// partial line of its own nearly every time (the y-records streaming through evict it before its neighbours arrive).
// So a lane keeps its last eight t values in LDS and writes them together: WRITE_SIZE 58.2 GB, 37.5 -> 35.4 ms on one
// box, 37.4 -> 37.1 ms in an interleaved A/B on another: the bytes are saved, the clock barely answers.
// Streaming (nontemporal) stores for the points: 3x slower (118 ms) -- L2 is what merges a point's four 16-byte stores.
// (Also built and measured: parking the 64-byte y-points in LDS so that 4 / 8 / 16 neighbouring lanes write 64 / 128 /
// 256 contiguous bytes of ONE track -- TCP_TCC_WRITE_REQ 3.59 G -> 1.44 G per launch, the address unit's stall cycles a
// quarter -- 38.7 / 36.6 / 36.4 ms: the request count was never the limit.  Not kept.)
__global__ void __launch_bounds__(64, LT_DENSE_WAVES) k_dense_tracks(DenseConsts k, const double *__restrict__ state0, DenseOut o,
                                                                     unsigned long long *__restrict__ head,
                                                                     const int32_t *__restrict__ perm)
{
    __shared__ double tring[64 * 9]; // eight t values per lane (stride 9: the lanes of an LDS access fall in different banks)
    const int lane = (int)threadIdx.x;
    // t values are parked only in the length-binned launch, where the 64 lanes accept their steps in lockstep and write
    // their eight values out in the same iteration; in caller order the lanes fill up at different times, every one of
    // them then pays its own eight store instructions, and the launch is 8 % SLOWER with the ring than without (A/B on one box)
    const bool ring = perm != nullptr;
    int64_t block = blockIdx.x;
    for (;;) {
    if (head) {
        unsigned long long w = 0;
        if (threadIdx.x == 0) w = atomicAdd(head, 1ull);
        block = (int64_t)(((uint64_t)__builtin_amdgcn_readfirstlane((uint32_t)(w >> 32)) << 32) |
                          (uint64_t)__builtin_amdgcn_readfirstlane((uint32_t)w));
    }
    if (block * 64 >= k.n) return;
    const int64_t slot_i = block * 64 + threadIdx.x;
    if (slot_i < k.n) {
    const int32_t i = perm ? perm[slot_i] : (int32_t)slot_i; // (n < 2^31, checked by the host: one register, not two)
    // Dormand-Prince tableau (Dormand & Prince 1980) and Shampine's dense-output matrix
    constexpr double A21 = 1.0 / 5, A31 = 3.0 / 40, A32 = 9.0 / 40, A41 = 44.0 / 45, A42 = -56.0 / 15, A43 = 32.0 / 9,
                     A51 = 19372.0 / 6561, A52 = -25360.0 / 2187, A53 = 64448.0 / 6561, A54 = -212.0 / 729,
                     A61 = 9017.0 / 3168, A62 = -355.0 / 33, A63 = 46732.0 / 5247, A64 = 49.0 / 176, A65 = -5103.0 / 18656;
    constexpr double B1 = 35.0 / 384, B3 = 500.0 / 1113, B4 = 125.0 / 192, B5 = -2187.0 / 6784, B6 = 11.0 / 84;
    constexpr double E1 = -71.0 / 57600, E3 = 71.0 / 16695, E4 = -71.0 / 1920, E5 = 17253.0 / 339200, E6 = -22.0 / 525, E7 = 1.0 / 40;
    constexpr double P[7][4] = {
        {1, -8048581381.0 / 2820520608, 8663915743.0 / 2820520608, -12715105075.0 / 11282082432},
        {0, 0, 0, 0},
        {0, 131558114200.0 / 32700410799, -68118460800.0 / 10900136933, 87487479700.0 / 32700410799},
        {0, -1754552775.0 / 470086768, 14199869525.0 / 1410260304, -10690763975.0 / 1880347072},
        {0, 127303824393.0 / 49829197408, -318862633887.0 / 49829197408, 701980252875.0 / 199316789632},
        {0, -282668133.0 / 205662961, 2019193451.0 / 616988883, -1453857185.0 / 822651844},
        {0, 40617522.0 / 29380423, -110615467.0 / 29380423, 69997945.0 / 29380423}};

    double y[8], f[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) y[c] = state0[(int64_t)i * 8 + c];
    const double r_in = k.r_in, r_out = k.r_out < 0 ? 2.0 * y[1] : k.r_out;
    double t = 0.0;
    int32_t n_pts = 0, nfev = 0, nt = 0;
    int status = 0;
    const int64_t rec0 = (int64_t)i * k.max_points; // this track's first record slot
    auto flush_t = [&](int32_t first) { // the parked t values belong to the points first, first + 1, ...
        for (int q = 0; q < nt; ++q) o.t[rec0 + first + q] = tring[lane * 9 + q];
        nt = 0;
    };
    auto push = [&](double tt, const double *yy) {
        const int64_t slot = rec0 + (n_pts < k.max_points ? n_pts : k.max_points - 1);
#ifndef LT_DENSE_NOSTORE // (diagnostic builds only: the kernel without its record stores)
        if (ring && n_pts < k.max_points - 1) {
            tring[lane * 9 + nt] = tt;
            if (++nt == 8) flush_t(n_pts - 7);
        } else { // the record's last slot is rewritten by every further point (it ends up holding the final one)
            flush_t((int32_t)k.max_points - 1 - nt);
            o.t[slot] = tt;
        }
        double2 *rec = reinterpret_cast<double2 *>(o.y + slot * 8); // 64-byte aligned (hipMalloc'd base, 64 B per point)
#pragma unroll
        for (int c = 0; c < 4; ++c) rec[c] = make_double2(yy[2 * c], yy[2 * c + 1]);
#else
        if (tt == -12345.678) o.t[slot] = yy[0] + yy[1] + yy[2] + yy[3] + yy[4] + yy[5] + yy[6] + yy[7];
#endif
        ++n_pts;
    };
    push(t, y);
    double s0, c0, ss, cc, sn, cn; // sin / cos of y[2], carried with f (FSAL); stage values; candidate next state's
    M<double>::sincos(y[2], s0, c0);
    rhs8_sc(k, y, s0, c0, f); ++nfev;
    double h_abs;
    { // initial step (Hairer, Norsett, Wanner II.4), error-estimator order 4
        double v[8], sc[8], y1[8], f1[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) { sc[c] = k.atol + fabs(y[c]) * k.rtol; v[c] = y[c] / sc[c]; }
        double d0 = rms8(v);
#pragma unroll
        for (int c = 0; c < 8; ++c) v[c] = f[c] / sc[c];
        double d1 = rms8(v);
        double h0 = (d0 < 1e-5 || d1 < 1e-5) ? 1e-6 : 0.01 * d0 / d1;
        h0 = fmin(h0, k.lambda_max);
#pragma unroll
        for (int c = 0; c < 8; ++c) y1[c] = y[c] + h0 * f[c];
        rhs8(k, y1, f1); ++nfev;
#pragma unroll
        for (int c = 0; c < 8; ++c) v[c] = (f1[c] - f[c]) / sc[c];
        double d2 = rms8(v) / h0;
        double h1 = (d1 <= 1e-15 && d2 <= 1e-15) ? fmax(1e-6, h0 * 1e-3) : pow(0.01 / fmax(d1, d2), 0.2);
        h_abs = fmin(fmin(100.0 * h0, h1), fmin(k.lambda_max, k.max_step));
    }
    double g_in = y[1] - r_in, g_out = y[1] - r_out;
    int32_t attempts = 0;
    // The six stages of one attempt from (y, f, s0, c0) with step hh: k3..k7, the candidate yn and its trigonometry.
    auto stages = [&](double hh, double *yn, double *k3, double *k4, double *k5, double *k6, double *k7) {
        double k2[8], tmp[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) tmp[c] = dense_cyclic(c) ? y[c] : y[c] + (f[c] * A21) * hh;
        rhs8_near(k, tmp, y[2], s0, c0, k2, ss, cc);
#pragma unroll
        for (int c = 0; c < 8; ++c) tmp[c] = dense_cyclic(c) ? y[c] : y[c] + (f[c] * A31 + k2[c] * A32) * hh;
        rhs8_near(k, tmp, y[2], s0, c0, k3, ss, cc);
#pragma unroll
        for (int c = 0; c < 8; ++c) tmp[c] = dense_cyclic(c) ? y[c] : y[c] + (f[c] * A41 + k2[c] * A42 + k3[c] * A43) * hh;
        rhs8_near(k, tmp, y[2], s0, c0, k4, ss, cc);
#pragma unroll
        for (int c = 0; c < 8; ++c) tmp[c] = dense_cyclic(c) ? y[c] : y[c] + (f[c] * A51 + k2[c] * A52 + k3[c] * A53 + k4[c] * A54) * hh;
        rhs8_near(k, tmp, y[2], s0, c0, k5, ss, cc);
#pragma unroll
        for (int c = 0; c < 8; ++c)
            tmp[c] = dense_cyclic(c) ? y[c] : y[c] + (f[c] * A61 + k2[c] * A62 + k3[c] * A63 + k4[c] * A64 + k5[c] * A65) * hh;
        rhs8_near(k, tmp, y[2], s0, c0, k6, ss, cc);
#pragma unroll
        for (int c = 0; c < 8; ++c) yn[c] = dense_cyclic(c) ? y[c] : y[c] + hh * (f[c] * B1 + k3[c] * B3 + k4[c] * B4 + k5[c] * B5 + k6[c] * B6);
        rhs8_near(k, yn, y[2], s0, c0, k7, sn, cn);
    };
    double h = 0, t_new = 0;
    double min_step = 0;
    bool new_step = true, rejected = false, hit_in = false, hit_out = false;
    while (t != k.lambda_max) {
        if (new_step) {
            min_step = 10.0 * fabs(nextafter(t, INFINITY) - t);
            h_abs = h_abs > k.max_step ? k.max_step : (h_abs < min_step ? min_step : h_abs);
            rejected = false;
            new_step = false;
        }
        if (h_abs < min_step || ++attempts > k.max_attempts) { status = attempts > k.max_attempts ? -2 : -1; break; }
        t_new = t + h_abs;
        if (t_new - k.lambda_max > 0) t_new = k.lambda_max;
        h = t_new - t;
        h_abs = fabs(h);
        double yn[8], k3[8], k4[8], k5[8], k6[8], k7[8];
        stages(h, yn, k3, k4, k5, k6, k7);
        nfev += 6;
        double e[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            if (dense_cyclic(c)) { e[c] = 0.0; continue; }
            double ec = (f[c] * E1 + k3[c] * E3 + k4[c] * E4 + k5[c] * E5 + k6[c] * E6 + k7[c] * E7) * h;
            e[c] = ec * M<double>::rcp_pos(k.atol + fmax(fabs(y[c]), fabs(yn[c])) * k.rtol); // (>= atol > 0)
        }
        const double err = rms8(e);
        if (!(err < 1.0)) { // rejected: this lane retries in the next iteration with a smaller step
            h_abs *= fmax(0.2, 0.9 * pow_m02(err)); // a NaN norm shrinks by 0.2, as in scipy (max(0.2, nan))
            rejected = true;
            continue;
        }
        double factor = (err == 0.0) ? 10.0 : fmin(10.0, 0.9 * pow_m02(err));
        if (rejected) factor = fmin(1.0, factor);
        h_abs *= factor;
        const double gn_in = yn[1] - r_in, gn_out = yn[1] - r_out;
        hit_in = g_in >= 0 && gn_in <= 0;
        hit_out = g_out <= 0 && gn_out >= 0;
        // terminal event inside this step: the lane leaves with (t, y, f, s0, c0, h) of the step's START; the event is
        // located below, once for the whole wave
        if (hit_in || hit_out) break;
        g_in = gn_in; g_out = gn_out;
        t = t_new;
#pragma unroll
        for (int c = 0; c < 8; ++c) { y[c] = yn[c]; f[c] = k7[c]; }
        s0 = sn; c0 = cn;
        push(t, y);
        new_step = true;
    }
    if (hit_in || hit_out) {
        // The stages of the step that contains the event are formed AGAIN here (same inputs, same operations, same
        // bits: one extra attempt per wave) rather than carried out of the loop -- 40 live float64 values would
        // cost the loop its second wave per SIMD.
        double yn[8], k3[8], k4[8], k5[8], k6[8], k7[8];
        stages(h, yn, k3, k4, k5, k6, k7);
        // dense output of that step, y(t) = y + h Q [x, x^2, x^3, x^4] with Q = K^T P: only the radius row is
        // needed to locate the event; the other rows are formed one at a time afterwards (registers)
        auto q_row = [&](int c, double *q) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                q[j] = f[c] * P[0][j] + k3[c] * P[2][j] + k4[c] * P[3][j] + k5[c] * P[4][j] + k6[c] * P[5][j] + k7[c] * P[6][j];
        };
        double Qr[4];
        q_row(1, Qr);
        // (one Brent call serves both kinds of lanes -- each with its own radius -- when no lane crossed both radii
        // in the same step, which is the rule: the two calls used to run one after the other)
        double root_in = 0, root_out = 0;
        if (!__ballot(hit_in && hit_out)) {
            const double root = brent_root(y[1], t, h, Qr, hit_in ? r_in : r_out, t, t_new);
            root_in = root_out = root;
        } else {
            if (hit_in) root_in = brent_root(y[1], t, h, Qr, r_in, t, t_new);
            if (hit_out) root_out = brent_root(y[1], t, h, Qr, r_out, t, t_new);
        }
        const bool take_in = hit_in && (!hit_out || root_in <= root_out);
        const double te = take_in ? root_in : root_out;
        const double x = (te - t) / h, x2 = x * x, x3 = x2 * x, x4 = x3 * x;
        double ye[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            double q[4];
            q_row(c, q);
            ye[c] = y[c] + h * (q[0] * x + q[1] * x2 + q[2] * x3 + q[3] * x4);
        }
        push(te, ye);
        status = take_in ? 1 : 2;
        // One value past the record: where the step that holds the event would have ended.  solve_ivp's continuous solution
        // (`solution.sol`) interpolates the last stretch with the dense output of that WHOLE step; with this the host can
        // rebuild it (geodesic_tracer.Track.sol).  The slot is outside the track's `count` points and only written when
        // the record has room.
        if (n_pts < k.max_points) o.t[rec0 + n_pts] = t_new;
    }
    flush_t((n_pts < k.max_points - 1 ? n_pts : (int32_t)k.max_points - 1) - nt);
    o.count[i] = n_pts;
    o.status[i] = (int8_t)status;
    o.nfev[i] = nfev;
    }
    if (!head) return;
    }
}

// ---- length-binned launch: predictor pass + windowed counting sort ------------------------------------------------
// Why: one track per lane, 64 consecutive tracks per wavefront, attempts per track 126 ... 232 (5th ... 95th
// percentile, 586 at most) bound the lane utilisation of k_dense_tracks at 0.63 (measured 0.596,
// profiles/r02_dense_tracks.txt) -- every wave waits for its longest track.  Sorted by length the bound is 1.
// The length is not known before the track is integrated, but it can be PREDICTED from a cheap integration of the same
// equations: the same Dormand-Prince pair at a loose tolerance (rtol_l, atol_l = rtol_l / 100 -- the ratio of the real
// pass, so that every component's error scale shrinks by the same factor) with a free step size.  An accepted loose
// step of length h with error norm err tells how many steps the REAL pass takes across it: the local error of the
// pair goes as h^5, so the real pass (tolerance ratio q = rtol / rtol_l, safety 0.9) takes steps of
// h_t = 0.9 h (q / err)^(1/5) there, capped by max_step:  n = max(h / max_step, err^(1/5) / (0.9 q^(1/5))).
// Summed over the track (the step that holds the terminal event counted up to the event, by linear interpolation in r):
// correlation 0.98 with the real attempt count, utilisation bound 0.91 when sorted by it, at 16 % of the real pass's
// attempts for rtol_l = 1e-4 (CPU twin: oracle/lt_oracle_dense.c lto_dense_step_log, tests/test_oracle_golden.py).
// The prediction only orders the launch; no output value depends on it.
constexpr int DENSE_KEY_BINS = 2048; // one bin per predicted attempt, the last one open-ended

template <typename T> struct PredictConsts {
    T M, a, a2, r_zero;
    int floor_sin2;
    T lambda_max, r_in, r_out; // r_out < 0: twice the track's start radius
    T rtol, atol;              // of the LOOSE pass
    T inv_step, inv_kappa;     // 1 / max_step of the real pass;  1 / (0.9 (rtol_real / rtol_loose)^(1/5))
    int64_t n;
    int32_t max_attempts;
    int32_t refill_min, chunk; // hand-out of tracks to idle lanes (k_dense_predict)
};

// the 8-D right-hand side of rhs8_sc in the predictor's arithmetic (float32 by default: a tolerance of 1e-4 is far above
// its rounding, and a float32 instruction issues at twice the rate of a float64 one)
template <typename T> __device__ __forceinline__ void rhs8_t(const PredictConsts<T> &k, const T *y, T *d)
{
    const T r = y[1], E = -y[4], pr = y[5], pth = y[6], L = y[7];
    if (r <= k.r_zero) {
#pragma unroll
        for (int i = 0; i < 8; ++i) d[i] = T(0);
        return;
    }
    T s, c;
    M<T>::sincos(y[2], s, c);
    T s2 = s * s;
    s2 = M<T>::max(s2, T(1e-15)); // (the Kerr form has no floor; the predictor only needs a finite number)
    const T r2 = r * r, ra = r2 + k.a2;
    const T Sigma = r2 + k.a2 * c * c, Delta = ra - T(2) * k.M * r;
    const T SD = Sigma * Delta, t = M<T>::rcp(SD * s2);
    const T iS = (Delta * s2) * t, iD = (Sigma * s2) * t, is2 = SD * t;
    const T P = E * ra - k.a * L, D = L - k.a * E * s2;
    const T PD = P * iD, Ds = D * is2;
    const T F = Delta * pr * pr + pth * pth + D * Ds - P * PD;
    const T H2 = F * iS;
    d[0] = (k.a * D + ra * PD) * iS;
    d[1] = Delta * pr * iS;
    d[2] = pth * iS;
    d[3] = (Ds + k.a * PD) * iS;
    d[4] = T(0);
    d[5] = -iS * ((r - k.M) * (pr * pr + PD * PD) - r * (T(2) * E * PD + H2));
    d[6] = iS * (s * c) * (Ds * (T(2) * k.a * E + Ds) - H2 * k.a2);
    d[7] = T(0);
}

template <typename T> __device__ __forceinline__ T rms8_t(const T *x)
{
    T s = T(0);
#pragma unroll
    for (int i = 0; i < 8; ++i) s += x[i] * x[i];
    return (T)__builtin_sqrtf((float)s) * T(0.35355339059327373);
}

// x^p for x > 0 through the hardware log2 / exp2 (relative error ~1e-6: the step controller and the prediction need no more)
__device__ __forceinline__ float fast_pow(float x, float p) { return __builtin_amdgcn_exp2f(p * __builtin_amdgcn_logf(x)); }

// One track per lane; writes the track's key (predicted attempts, clamped).
// The loose pass's lengths spread as widely as the real pass's -- a wavefront of 64 consecutive tracks that waits for its
// longest ran at a lane utilisation of 0.48 -- and there is no predictor for the predictor.  There does not have to be: a
// key is one 16-bit store at the track's index, so a lane whose track has ended takes the next track of the launch.
// A wave takes `chunk` track numbers from `head` at a time (DENSE_PRED_CHUNK) and deals them out to its idle lanes (each
// takes next + its rank among them) whenever at least `refill_min` lanes are idle (DENSE_REFILL_MIN): the set-up of a
// track (Hairer's initial step: two right-hand sides) then runs for that many lanes at once, and an attempt for more
// than 64 - refill_min.  Measured at 4 M tracks (tools/scratch/pred_sweep.sh): 3.60 ms block per wave -> 2.68 ms, lane
// utilisation 0.48 -> 0.74 (what is left: the set-up at partial occupancy, the idle lanes below the threshold, the end of
// every wave's life); refill_min 1 / 2 / 4 / 8 / 16 / 24: 3.55 / 3.33 / 2.96 / 2.68 / 2.70 / 2.83 ms; chunk 32 / 64 / 128:
// 2.80 / 2.68 / 2.76 ms.  (One atomic per REFILL instead of per chunk -- 420 k atomics on one address -- took 5.6 ms: the
// memory side serialises same-address atomics at ~6 ns each, and every wave waits for its own.)
// head == nullptr: the workgroup's own block of 64 tracks and nothing after it.
// A track's arithmetic does not depend on the lane or the company it runs in: the keys are those of one block per wave.
constexpr int DENSE_REFILL_MIN = 8;
constexpr int DENSE_PRED_CHUNK = 64;

template <typename T>
__global__ void __launch_bounds__(64) k_dense_predict(PredictConsts<T> k, const double *__restrict__ state0,
                                                      uint16_t *__restrict__ key, unsigned long long *__restrict__ head)
{
    constexpr T A21 = T(1.0 / 5), A31 = T(3.0 / 40), A32 = T(9.0 / 40), A41 = T(44.0 / 45), A42 = T(-56.0 / 15), A43 = T(32.0 / 9),
                A51 = T(19372.0 / 6561), A52 = T(-25360.0 / 2187), A53 = T(64448.0 / 6561), A54 = T(-212.0 / 729),
                A61 = T(9017.0 / 3168), A62 = T(-355.0 / 33), A63 = T(46732.0 / 5247), A64 = T(49.0 / 176), A65 = T(-5103.0 / 18656);
    constexpr T B1 = T(35.0 / 384), B3 = T(500.0 / 1113), B4 = T(125.0 / 192), B5 = T(-2187.0 / 6784), B6 = T(11.0 / 84);
    constexpr T E1 = T(-71.0 / 57600), E3 = T(71.0 / 16695), E4 = T(-71.0 / 1920), E5 = T(17253.0 / 339200), E6 = T(-22.0 / 525), E7 = T(1.0 / 40);
    const int lane = (int)threadIdx.x;
    const uint64_t below = (1ull << lane) - 1ull;
    // the lane's track
    T y[8], f[8];
    T h = T(0), t = T(0), pred = T(0), g_in = T(0), g_out = T(0), r_in = T(0), r_out = T(0);
    int64_t i = 0;
    int32_t attempts = 0;
    bool rejected = false, have = false;
    bool drained = false;          // (wave-uniform) nothing left to hand out
    int64_t next = 0, next_end = 0; // (wave-uniform) the track numbers the wave holds and has not started yet
#pragma unroll
    for (int c = 0; c < 8; ++c) y[c] = f[c] = T(0);
    for (;;) {
        const uint64_t idle = __ballot(!have);
        const int n_idle = __popcll(idle);
        if (!drained && (n_idle >= k.refill_min || n_idle == 64)) {
            if (next == next_end) { // the wave's chunk of track numbers is used up: take the next one
                if (head) {
                    unsigned long long w = 0;
                    if (lane == 0) w = atomicAdd(head, (unsigned long long)k.chunk);
                    next = (int64_t)(((uint64_t)__builtin_amdgcn_readfirstlane((uint32_t)(w >> 32)) << 32) |
                                     (uint64_t)__builtin_amdgcn_readfirstlane((uint32_t)w));
                    next_end = next + k.chunk < k.n ? next + k.chunk : k.n;
                } else if (next_end == 0) {
                    next = (int64_t)blockIdx.x * 64;
                    next_end = next + 64 < k.n ? next + 64 : k.n;
                } else {
                    next = next_end = k.n;
                }
                if (next >= k.n) { drained = true; next = next_end = k.n; }
            }
            const int64_t cand = next + __popcll(idle & below);
            const int64_t avail = next_end - next;
            const bool take = !have && cand < next_end;
            next += n_idle < avail ? n_idle : avail;
            if (take) {
                i = cand;
                have = true;
#pragma unroll
                for (int c = 0; c < 8; ++c) y[c] = (T)state0[i * 8 + c];
                r_in = k.r_in;
                r_out = k.r_out < T(0) ? T(2) * y[1] : k.r_out;
                rhs8_t(k, y, f);
                { // Hairer's initial step, as the real pass (at the loose tolerance)
                    T v[8], sc[8], y1[8], f1[8];
#pragma unroll
                    for (int c = 0; c < 8; ++c) { sc[c] = k.atol + M<T>::abs(y[c]) * k.rtol; v[c] = y[c] / sc[c]; }
                    T d0 = rms8_t(v);
#pragma unroll
                    for (int c = 0; c < 8; ++c) v[c] = f[c] / sc[c];
                    T d1 = rms8_t(v);
                    T h0 = (d0 < T(1e-5) || d1 < T(1e-5)) ? T(1e-6) : T(0.01) * d0 / d1;
                    h0 = M<T>::min(h0, k.lambda_max);
#pragma unroll
                    for (int c = 0; c < 8; ++c) y1[c] = y[c] + h0 * f[c];
                    rhs8_t(k, y1, f1);
#pragma unroll
                    for (int c = 0; c < 8; ++c) v[c] = (f1[c] - f[c]) / sc[c];
                    T d2 = rms8_t(v) / h0;
                    T dm = M<T>::max(d1, d2);
                    T h1 = dm <= T(1e-15) ? M<T>::max(T(1e-6), h0 * T(1e-3)) : (T)fast_pow(0.01f / (float)dm, 0.2f);
                    h = M<T>::min(M<T>::min(T(100) * h0, h1), k.lambda_max);
                }
                t = T(0);
                pred = T(0);
                g_in = y[1] - r_in;
                g_out = y[1] - r_out;
                rejected = false;
                attempts = 0;
            }
        }
        if (__ballot(have) == 0ull) {
            if (drained) break;
            continue;
        }
        if (have) { // one attempt
            bool ended = false, finished = false; // finished: the track is over; ended: and the loose pass saw its end
            if (attempts >= k.max_attempts) {
                finished = true;
            } else if (!(t < k.lambda_max) || !(h > T(1e-6) * (T(1) + t))) { // (a collapsed step size: whatever was predicted so far)
                finished = ended = true;
            } else {
                ++attempts;
                T hh = M<T>::min(h, k.lambda_max - t);
                T k2[8], k3[8], k4[8], k5[8], k6[8], k7[8], tmp[8], yn[8];
#pragma unroll
                for (int c = 0; c < 8; ++c) tmp[c] = dense_cyclic(c) ? y[c] : y[c] + (f[c] * A21) * hh;
                rhs8_t(k, tmp, k2);
#pragma unroll
                for (int c = 0; c < 8; ++c) tmp[c] = dense_cyclic(c) ? y[c] : y[c] + (f[c] * A31 + k2[c] * A32) * hh;
                rhs8_t(k, tmp, k3);
#pragma unroll
                for (int c = 0; c < 8; ++c) tmp[c] = dense_cyclic(c) ? y[c] : y[c] + (f[c] * A41 + k2[c] * A42 + k3[c] * A43) * hh;
                rhs8_t(k, tmp, k4);
#pragma unroll
                for (int c = 0; c < 8; ++c) tmp[c] = dense_cyclic(c) ? y[c] : y[c] + (f[c] * A51 + k2[c] * A52 + k3[c] * A53 + k4[c] * A54) * hh;
                rhs8_t(k, tmp, k5);
#pragma unroll
                for (int c = 0; c < 8; ++c) tmp[c] = dense_cyclic(c) ? y[c] : y[c] + (f[c] * A61 + k2[c] * A62 + k3[c] * A63 + k4[c] * A64 + k5[c] * A65) * hh;
                rhs8_t(k, tmp, k6);
#pragma unroll
                for (int c = 0; c < 8; ++c) yn[c] = dense_cyclic(c) ? y[c] : y[c] + hh * (f[c] * B1 + k3[c] * B3 + k4[c] * B4 + k5[c] * B5 + k6[c] * B6);
                rhs8_t(k, yn, k7);
                T e[8];
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    if (dense_cyclic(c)) { e[c] = T(0); continue; }
                    T ec = (f[c] * E1 + k3[c] * E3 + k4[c] * E4 + k5[c] * E5 + k6[c] * E6 + k7[c] * E7) * hh;
                    e[c] = ec * M<T>::rcp(k.atol + M<T>::max(M<T>::abs(y[c]), M<T>::abs(yn[c])) * k.rtol);
                }
                const T err = rms8_t(e);
                if (!(err < T(1))) { // rejected (or NaN): shrink, try again
                    T fac = err == err ? (T)(0.9f * fast_pow((float)err, -0.2f)) : T(0.2);
                    h = hh * M<T>::max(T(0.2), fac);
                    rejected = true;
                } else {
                    T fac = err > T(1e-12) ? M<T>::min(T(10), (T)(0.9f * fast_pow((float)err, -0.2f))) : T(10);
                    if (rejected) fac = M<T>::min(T(1), fac);
                    rejected = false;
                    h = hh * fac;
                    const T gn_in = yn[1] - r_in, gn_out = yn[1] - r_out;
                    const bool hit_in = g_in >= T(0) && gn_in <= T(0), hit_out = g_out <= T(0) && gn_out >= T(0);
                    T used = hh;
                    if (hit_in || hit_out) { // the part of the step before the event, by linear interpolation in r
                        T g0 = hit_in ? g_in : g_out, g1 = hit_in ? gn_in : gn_out;
                        T fr = g0 != g1 ? g0 / (g0 - g1) : T(1);
                        used = hh * M<T>::min(M<T>::max(fr, T(0)), T(1));
                    }
                    // steps of the real pass across this one: held by max_step, or by the tolerance
                    const T by_tol = err > T(1e-12) ? (T)fast_pow((float)err, 0.2f) * k.inv_kappa * (used / hh) : T(0);
                    pred += M<T>::max(used * k.inv_step, by_tol);
                    if (hit_in || hit_out) finished = ended = true;
                    g_in = gn_in; g_out = gn_out;
                    t += hh;
#pragma unroll
                    for (int c = 0; c < 8; ++c) { y[c] = yn[c]; f[c] = k7[c]; }
                }
            }
            if (finished) {
                // a track the loose pass did not finish within its attempt budget (a photon that orbits the hole for long: a few
                // per cent of a fan of rays, several times the typical effort) is long: it goes first in its window
                if (!ended) pred = T(DENSE_KEY_BINS);
                int kk = pred == pred ? (int)(float)pred : 0;
                kk = kk < 0 ? 0 : (kk > DENSE_KEY_BINS - 1 ? DENSE_KEY_BINS - 1 : kk);
                key[i] = (uint16_t)kk;
                have = false;
            }
        }
    }
}

#ifndef LT_KERNEL_TEMPLATES_ONLY
// perm: inside every WINDOW of `window` consecutive tracks, the tracks by descending key; windows stay in caller order.
// Why windows and not one global order: a lane appends to ITS track's record, so the 64 lanes of a wavefront write 64
// different places for the whole life of the wave.  In caller order those are 64 neighbouring records (one or two
// pages); in a global longest-first order they are 64 random places of a record buffer of tens of GB -- measured at 4 M
// tracks (67 GB of records): the stores, 1.9 ms of the launch in caller order, cost 8.5 ms, all that the sort had won
// (address translation: every store instruction then touches 64 pages no other wave shares).  Within a window of 2 048
// tracks (29 MB of records) a wave's lanes stay within a few pages, and sorting inside windows keeps nearly all of the
// gain: the utilisation bound of one track per lane is 0.90 for windows of 2 048 against 0.92 for the global order.
// One workgroup per window: histogram of the keys in LDS, prefix over the bins from the top, scatter -- no global
// atomics, no second kernel.  The order inside a bin is whatever the LDS atomics made it; it does not matter.
__global__ void __launch_bounds__(256) k_dense_window_sort(const uint16_t *__restrict__ key, int32_t *__restrict__ perm,
                                                           int64_t n, int window)
{
    __shared__ uint32_t cnt[DENSE_KEY_BINS];
    __shared__ uint32_t part[256];
    const int t = threadIdx.x;
    for (int b = t; b < DENSE_KEY_BINS; b += 256) cnt[b] = 0;
    __syncthreads();
    const int64_t base = (int64_t)blockIdx.x * window;
    const int64_t end = base + window < n ? base + window : n;
    for (int64_t i = base + t; i < end; i += 256) atomicAdd(&cnt[key[i]], 1u);
    __syncthreads();
    // exclusive prefix over the bins from the top: thread t owns the bins [hi - PER + 1, hi], hi = BINS - 1 - t * PER
    constexpr int PER = DENSE_KEY_BINS / 256;
    const int hi = DENSE_KEY_BINS - 1 - t * PER;
    uint32_t own[PER], sum = 0;
#pragma unroll
    for (int j = 0; j < PER; ++j) { own[j] = cnt[hi - j]; sum += own[j]; }
    part[t] = sum;
    __syncthreads();
    for (int off = 1; off < 256; off <<= 1) {
        uint32_t v = t >= off ? part[t - off] : 0u;
        __syncthreads();
        part[t] += v;
        __syncthreads();
    }
    uint32_t run = part[t] - sum;
#pragma unroll
    for (int j = 0; j < PER; ++j) { cnt[hi - j] = run; run += own[j]; }
    __syncthreads();
    for (int64_t i = base + t; i < end; i += 256) {
        const uint32_t pos = atomicAdd(&cnt[key[i]], 1u);
        perm[base + pos] = (int32_t)i; // pos < end - base: the bins hold exactly the window's tracks
    }
}
#endif // LT_KERNEL_TEMPLATES_ONLY

// probe of rhs8 for the parity tests
__global__ void __launch_bounds__(64) k_rhs8_probe(DenseConsts k, const double *__restrict__ states, double *__restrict__ out)
{
    const int64_t i = (int64_t)blockIdx.x * 64 + threadIdx.x;
    if (i >= k.n) return;
    double y[8], d[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) y[c] = states[i * 8 + c];
    rhs8(k, y, d);
#pragma unroll
    for (int c = 0; c < 8; ++c) out[i * 8 + c] = d[c];
}

} // namespace lt
