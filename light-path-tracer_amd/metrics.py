"""Spacetime-metric plugin surface, MI355X backend.

Same public surface as the reference's metrics.py (class Metric / Schwarzschild /
Kerr, reference metrics.py:682-1132): constructor arguments, method names,
argument meaning, return conventions and error behaviour are kept, so code
written against the reference (image_lens.py, main.py, user plugins) runs
unchanged.  What differs is where rays are integrated: `trace_ray` and
`trace_rays_batch` call the hand-written gfx950 kernels in libltrace_hip.so
through ctypes (ltrace.py).  There is no numba / CPU tracer in this package: with
no GPU, tracing raises ltrace.LtraceError.

Host-side scalar physics that is not on the per-pixel path (critical angles,
impact parameters, the 8-D Hamiltonian right-hand side consumed by
scipy.solve_ivp in geodesic_tracer.py) is plain numpy.

Extra, backend-only knobs (keyword arguments; left out, the reference's own
arithmetic is used -- float64, DP45 for Kerr): integrator = 'rk4' | 'dp45' |
'dp45_exact', precision = 32 | 64, schedule = 'direct' | 'queue'.
"""
from abc import ABC, abstractmethod

import numpy as np

import ltrace

# Backend defaults = the reference's own arithmetic, so that switching to this package changes where rays are integrated
# and nothing else: float64 everywhere (the reference has no float32 path), and for Kerr its production integrator, DP45
# (metrics.py:419-567), with the step controller evaluated in float64 as the reference writes it ('dp45_exact': the
# accept / reject sequence of the reference's own run, see INTEGRATION.md for what "exact" covers; 4 % over 'dp45',
# whose controller is float32).  The fast path is opt-in: integrator='rk4', precision=32 is the north-star kernel
# (reference metrics.py:570-658 in float32), about 1.7x faster per ray -- what bench.py's headline measures --
# and Schwarzschild(precision=32) the float32 orbit-equation tracer.
DEFAULT_KERR_INTEGRATOR = "dp45_exact"
DEFAULT_PRECISION = 64
DEFAULT_SCHEDULE = "direct"

_OUTCOME = {1: "escaped", -1: "captured", 0: "invalid"}


class Metric(ABC):
    """Base class for spacetime metrics (reference metrics.py:682-728).

    Public 8-D state: [t, r, theta, phi, p_t, p_r, p_theta, p_phi]."""

    is_spherically_symmetric = False

    @abstractmethod
    def geodesic_equations(self, lambda_, state):
        """Right-hand side of Hamilton's equations for the 8-D state."""

    @abstractmethod
    def initial_conditions(self, r_obs, alpha, theta=0.0, theta_obs=np.pi / 2):
        """Initial 8-D state of a photon seen at viewing angle alpha; None if there is none."""

    @abstractmethod
    def trace_ray(self, r_obs, alpha, theta=0.0, theta_obs=np.pi / 2, phi_max=50.0, axis_refine=False):
        """-> (final_alpha, n_half_orbits, 'escaped' | 'captured' | 'invalid')."""

    @abstractmethod
    def alpha_crit(self, r_obs, theta_obs=np.pi / 2):
        """Critical viewing angle in radians."""

    @abstractmethod
    def capture_radius(self):
        """Inner stopping radius of the integration."""

    def viewing_angle_to_impact_parameter(self, alpha, r_obs, theta_obs=np.pi / 2):
        raise NotImplementedError


def _single(fa, nh, st):
    s = int(st[0])
    if s == 0:
        return np.nan, 0, "invalid"
    if s == -1:
        return np.nan, int(nh[0]), "captured"
    return float(fa[0]), int(nh[0]), "escaped"


class Schwarzschild(Metric):
    """Non-rotating black hole of mass M (reference metrics.py:735-833)."""

    is_spherically_symmetric = True

    def __init__(self, M=1.0, precision=None):
        self.M = M
        self.R_S = 2 * M
        self.R_PHOTON = 3 * M
        self.B_CRIT = 3 * np.sqrt(3) * M
        self.precision = DEFAULT_PRECISION if precision is None else precision

    def _f(self, r):
        return 1 - self.R_S / r

    def capture_radius(self):
        return self.R_S * 1.01

    def alpha_crit(self, r_obs, theta_obs=np.pi / 2):
        return np.arcsin(np.clip(self.B_CRIT * np.sqrt(self._f(r_obs)) / r_obs, -1.0, 1.0))

    def viewing_angle_to_impact_parameter(self, alpha, r_obs, theta_obs=np.pi / 2):
        return r_obs * np.sin(alpha) / np.sqrt(self._f(r_obs))

    def geodesic_equations(self, lambda_, state):
        """8-D right-hand side (reference metrics.py:763-790)."""
        _t, r, th, _phi, p_t, p_r, p_th, p_phi = state
        if r <= self.R_S * 1.001:
            return [0.0] * 8
        f = self._f(r)
        s, c = np.sin(th), np.cos(th)
        s2 = max(s * s, 1e-15)
        half_rs_r2 = self.R_S / (2 * r * r)
        ang = p_th * p_th + p_phi * p_phi / s2
        return [-p_t / f,
                f * p_r,
                p_th / (r * r),
                p_phi / (r * r * s2),
                0.0,
                -half_rs_r2 * (p_t * p_t / (f * f)) - half_rs_r2 * p_r * p_r + ang / r**3,
                c * p_phi * p_phi / (r * r * s2 * s),
                0.0]

    def initial_conditions(self, r_obs, alpha, theta=0.0, theta_obs=np.pi / 2):
        """Equatorial photon launched inward (reference metrics.py:794-809)."""
        L = self.viewing_angle_to_impact_parameter(alpha, r_obs)
        f0 = self._f(r_obs)
        p_r_sq = (1.0 / f0 - L * L / (r_obs * r_obs)) / f0
        if p_r_sq < 0:
            return None
        return [0.0, r_obs, np.pi / 2, 0.0, -1.0, -np.sqrt(p_r_sq), 0.0, L]

    def trace_ray(self, r_obs, alpha, theta=0.0, theta_obs=np.pi / 2, phi_max=50.0, axis_refine=False):
        """One ray through the GPU orbit-equation tracer (reference metrics.py:817-829)."""
        fa = np.full(1, np.nan)
        nh = np.zeros(1, dtype=np.int64)
        st = np.zeros(1, dtype=np.int8)
        ltrace.trace_batch_schw(self.M, r_obs, np.array([alpha], dtype=np.float64), fa, nh,
                                phi_max=phi_max, h_max=0.05, precision=self.precision, out_status=st)
        return _single(fa, nh, st)

    def trace_rays_batch(self, r_obs, alphas, out_fa, out_w):
        """In place, like the reference (metrics.py:831-833): phi_max 50, h 0.05."""
        if not (out_fa.flags.c_contiguous and out_w.flags.c_contiguous):
            raise ValueError("out_fa / out_w must be C-contiguous")
        ltrace.trace_batch_schw(self.M, r_obs, alphas, out_fa, out_w, phi_max=50.0, h_max=0.05,
                                precision=self.precision)


class Kerr(Metric):
    """Kerr black hole in Boyer-Lindquist coordinates, |a| <= M (reference metrics.py:840-1132)."""

    is_spherically_symmetric = False

    def __init__(self, M=1.0, a=0.0, integrator=None, precision=None, schedule=None):
        if abs(a) > M:
            raise ValueError(f"|a|={abs(a)} exceeds M={M}")
        self.M = M
        self.a = a
        self.r_plus = M + np.sqrt(M**2 - a**2)
        self.integrator = DEFAULT_KERR_INTEGRATOR if integrator is None else integrator
        self.precision = DEFAULT_PRECISION if precision is None else precision
        if self.integrator in ("dp45", "dp45_exact"):
            self.precision = 64
        self.schedule = DEFAULT_SCHEDULE if schedule is None else schedule

    def _Sigma(self, r, th):
        return r**2 + self.a**2 * np.cos(th)**2

    def _Delta(self, r):
        return r**2 - 2 * self.M * r + self.a**2

    def capture_radius(self):
        return self.r_plus * 1.01

    # -- photon orbits (reference metrics.py:866-930) ------------------------------------------
    def _unstable_photon_r(self):
        M, a = self.M, self.a
        if a == 0:
            return 3 * M, 3 * M
        return (2 * M * (1 + np.cos(2 / 3 * np.arccos(-a / M))),
                2 * M * (1 + np.cos(2 / 3 * np.arccos(a / M))))

    def _xi_eta(self, r_ph):
        """Conserved (xi, eta) of the spherical photon orbit at radius r_ph (Bardeen)."""
        M, a = self.M, self.a
        Delta = self._Delta(r_ph)
        xi = (r_ph**2 + a**2) / a - 2 * r_ph * Delta / (a * (r_ph - M))
        eta = r_ph**3 / (a**2 * (r_ph - M)**2) * (4 * M * Delta - r_ph * (r_ph - M)**2)
        return xi, eta

    def _critical_impact_params(self):
        if self.a == 0:
            raise ValueError("_critical_impact_params undefined for a=0")
        return [self._xi_eta(r) for r in self._unstable_photon_r()]

    def alpha_crit(self, r_obs, theta_obs=np.pi / 2):
        """Conservative shadow envelope: the largest impact parameter over 50 sampled
        spherical photon orbits, floored by the Schwarzschild value."""
        M, a = self.M, self.a
        if a == 0:
            return np.arcsin(np.clip(3 * np.sqrt(3) * M * np.sqrt(1 - 2 * M / r_obs) / r_obs, -1.0, 1.0))
        r_pro, r_ret = self._unstable_photon_r()
        b2_max = 0.0
        for r_ph in np.linspace(r_pro, r_ret, 50):
            xi, eta = self._xi_eta(r_ph)
            b2_max = max(b2_max, xi**2 + max(eta, 0.0))
        b_crit = max(np.sqrt(b2_max), 3 * np.sqrt(3) * M)
        Delta, Sigma = self._Delta(r_obs), self._Sigma(r_obs, theta_obs)
        A = (r_obs**2 + a**2)**2 - a**2 * Delta * np.sin(theta_obs)**2
        return np.arcsin(np.clip(b_crit * np.sqrt(Sigma * Delta / A) / r_obs, -1.0, 1.0))

    def viewing_angle_to_impact_parameter(self, alpha, r_obs, theta_obs=np.pi / 2):
        if self.a == 0:
            return r_obs * np.sin(alpha) / np.sqrt(1 - 2 * self.M / r_obs)
        Delta, Sigma = self._Delta(r_obs), self._Sigma(r_obs, theta_obs)
        A = (r_obs**2 + self.a**2)**2 - self.a**2 * Delta * np.sin(theta_obs)**2
        return r_obs * np.sin(alpha) * np.sqrt(A / (Sigma * Delta))

    # -- 8-D Hamiltonian flow (reference metrics.py:946-1029) -----------------------------------
    def geodesic_equations(self, lambda_, state):
        """dx/dlambda = dH/dp, dp/dlambda = -dH/dx for H = g^{mu nu} p_mu p_nu / 2.

        Written from the separable form 2 Sigma H = Delta p_r^2 + p_th^2 + (L - a E s^2)^2 / s^2
        - ((r^2 + a^2) E - a L)^2 / Delta (E = -p_t, L = p_phi) -- the same form the GPU kernel
        inlines (csrc/lt_device.hpp) -- instead of differentiating each g^{mu nu} separately."""
        _t, r, th, _phi, p_t, p_r, p_th, L = state
        M, a = self.M, self.a
        if r <= self.r_plus * 1.001:
            return [0.0] * 8
        E = -p_t
        s, c = np.sin(th), np.cos(th)
        s2 = s * s
        Sigma = r * r + a * a * c * c
        Delta = r * r - 2 * M * r + a * a
        P = (r * r + a * a) * E - a * L
        K = L - a * E * s2
        F = Delta * p_r * p_r + p_th * p_th + K * K / s2 - P * P / Delta
        two_H = F / Sigma
        dDelta = 2 * r - 2 * M
        F_r = dDelta * p_r * p_r - (4 * r * E * P * Delta - P * P * dDelta) / (Delta * Delta)
        F_th = 2 * s * c * (a * a * E * E - L * L / (s2 * s2))
        return [(a * K + (r * r + a * a) * P / Delta) / Sigma,
                Delta * p_r / Sigma,
                p_th / Sigma,
                (K / s2 + a * P / Delta) / Sigma,
                0.0,
                -(F_r - two_H * 2 * r) / (2 * Sigma),
                -(F_th + two_H * 2 * a * a * s * c) / (2 * Sigma),
                0.0]

    def initial_conditions(self, r_obs, alpha, theta=0.0, theta_obs=np.pi / 2):
        """Photon at the observer; theta = screen azimuth (0 up, pi/2 right), theta_obs =
        inclination (reference metrics.py:1033-1109)."""
        M, a = self.M, self.a
        s_o, c_o = np.sin(theta_obs), np.cos(theta_obs)
        Sigma, Delta = self._Sigma(r_obs, theta_obs), self._Delta(r_obs)
        rho = r_obs * np.sin(alpha) * np.sqrt(Sigma) / np.sqrt(Delta)
        x_scr, y_scr = -rho * np.sin(theta), -rho * np.cos(theta)
        L = -x_scr * s_o
        Q = y_scr**2 + c_o**2 * (x_scr**2 - a**2)
        Theta = max(Q - c_o**2 * (L**2 / s_o**2 - a**2), 0.0)
        p_theta = (-1.0 if np.cos(theta) > 0 else 1.0) * np.sqrt(Theta)
        A = (r_obs**2 + a**2)**2 - a**2 * Delta * s_o**2
        g_tt = -A / (Sigma * Delta)
        g_tphi = -2 * M * a * r_obs / (Sigma * Delta)
        g_phiphi = (Delta - a**2 * s_o**2) / (Sigma * Delta * s_o**2)
        other = g_tt - 2 * g_tphi * L + p_theta**2 / Sigma + g_phiphi * L**2
        p_r = -np.sqrt(max(-other / (Delta / Sigma), 0.0))
        return [0.0, r_obs, theta_obs, 0.0, -1.0, p_r, p_theta, L]

    # -- ray tracing on the GPU ---------------------------------------------------------------------
    def _lambda_max(self, r_obs):
        return max(5000.0, 6.0 * r_obs)

    def trace_ray(self, r_obs, alpha, theta=0.0, theta_obs=np.pi / 2, phi_max=50.0, axis_refine=False):
        """One ray (reference metrics.py:1113-1126)."""
        fa = np.full(1, np.nan)
        nh = np.zeros(1, dtype=np.int64)
        st = np.zeros(1, dtype=np.int8)
        ltrace.trace_batch_kerr(self.M, self.a, r_obs, np.array([alpha], dtype=np.float64),
                                np.array([theta], dtype=np.float64), theta_obs, self._lambda_max(r_obs),
                                np.array([bool(axis_refine)]), fa, nh, integrator=self.integrator,
                                precision=self.precision, schedule=self.schedule, out_status=st)
        return _single(fa, nh, st)

    def trace_rays_batch(self, r_obs, alphas, thetas, theta_obs, axis_refines, out_fa, out_w):
        """In place, like the reference (metrics.py:1128-1132)."""
        if not (out_fa.flags.c_contiguous and out_w.flags.c_contiguous):
            raise ValueError("out_fa / out_w must be C-contiguous")
        ltrace.trace_batch_kerr(self.M, self.a, r_obs, alphas, thetas, theta_obs, self._lambda_max(r_obs),
                                axis_refines, out_fa, out_w, integrator=self.integrator,
                                precision=self.precision, schedule=self.schedule)
