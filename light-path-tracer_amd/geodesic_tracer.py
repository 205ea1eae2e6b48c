"""Dense single-ray integration and trajectory plots: the reference's geodesic_tracer.py surface
(`integrate_geodesic`, `trace_ray`, `plot_trajectories`, the demo under __main__).

This is the trajectory side of the metric plugin.  Two ways through it:

* `integrate_geodesic` / `trace_ray` -- one ray through scipy.solve_ivp on the metric's 8-D
  `geodesic_equations`, exactly what the reference does (its callers trace one to ten rays for a plot).
  Any Metric subclass works, including ones the GPU library knows nothing about.
* `integrate_geodesics` / `trace_rays` -- a whole batch on the GPU (`lt_integrate_dense`: the same
  Dormand-Prince 5(4) integrator, step controller, dense output and event location as solve_ivp's RK45,
  float64, one work-item per track).  Schwarzschild and Kerr only; raises without a GPU.
  `plot_trajectories(..., backend="hip")` draws from it.

Units: G = c = 1.
"""
import numpy as np
from scipy.integrate import solve_ivp

import ltrace
from metrics import Schwarzschild

# solve_ivp settings of the reference (geodesic_tracer.py:57-67)
IVP = dict(method="RK45", max_step=1.0, rtol=1e-8, atol=1e-10, dense_output=True)
DEMO_ANGLES = (0, 2, 4, 5, 5.5, 5.97, 6.5, 8, 10, 15)


def _radius_event(radius, direction):
    def event(_lam, y):
        return y[1] - radius
    event.terminal, event.direction = True, direction
    return event


def integrate_geodesic(metric, state0, lambda_max=1000.0, r_stop_inner=None, r_stop_outer=None):
    """8-D state0 -> (OdeSolution, 'captured' | 'escaped').  Stops at r_stop_inner (default the metric's
    capture radius, falling), r_stop_outer (default twice the start radius, rising) or lambda_max."""
    r_in = metric.capture_radius() if r_stop_inner is None else r_stop_inner
    r_out = 2.0 * state0[1] if r_stop_outer is None else r_stop_outer
    sol = solve_ivp(metric.geodesic_equations, [0, lambda_max], state0,
                    events=[_radius_event(r_in, -1), _radius_event(r_out, +1)], **IVP)
    return sol, ("captured" if sol.y[1, -1] <= 1.1 * r_in else "escaped")


def trace_ray(metric, r_obs, alpha, **kwargs):
    """Ray seen at viewing angle alpha from r_obs -> (solution, outcome), or (None, 'invalid')."""
    state0 = metric.initial_conditions(r_obs, alpha)
    return (None, "invalid") if state0 is None else integrate_geodesic(metric, state0, **kwargs)


# ---------------------------------------------------------------------------------------------
# the batched GPU path
# ---------------------------------------------------------------------------------------------
class Track:
    """One trajectory of a GPU batch, with the fields the reference's callers read off solve_ivp's result:
    t (n_points,), y (8, n_points) = rows t, r, theta, phi, p_t, p_r, p_theta, p_phi; status as solve_ivp
    (1 a radius event ended it, 0 lambda_max reached, -1 failed), nfev, success, and `sol(t)`, the continuous
    solution.  `event` names the event ('captured' / 'escaped' / None); `truncated` says the record did not fit
    `max_points` (then the last point is still the final one, and there is no dense output)."""
    __slots__ = ("t", "y", "status", "nfev", "event", "truncated", "success", "_metric", "_t_step_end", "_segments")

    def __init__(self, t, y, code, nfev, truncated, metric=None, t_step_end=None):
        self.t, self.y, self.nfev, self.truncated = t, y, int(nfev), bool(truncated)
        self.status = 1 if code in (ltrace.TRACK_CAPTURE_EVENT, ltrace.TRACK_ESCAPE_EVENT) else (0 if code == 0 else -1)
        self.event = {ltrace.TRACK_CAPTURE_EVENT: "captured", ltrace.TRACK_ESCAPE_EVENT: "escaped"}.get(code)
        self.success = self.status >= 0
        self._metric, self._t_step_end, self._segments = metric, t_step_end, {}

    # Dormand-Prince 5(4) tableau and Shampine's dense-output matrix (what scipy's RK45 uses)
    _C = np.array([0, 1 / 5, 3 / 10, 4 / 5, 8 / 9, 1])
    _A = [[], [1 / 5], [3 / 40, 9 / 40], [44 / 45, -56 / 15, 32 / 9], [19372 / 6561, -25360 / 2187, 64448 / 6561, -212 / 729],
          [9017 / 3168, -355 / 33, 46732 / 5247, 49 / 176, -5103 / 18656]]
    _B = np.array([35 / 384, 0, 500 / 1113, 125 / 192, -2187 / 6784, 11 / 84])
    _P = np.array([[1, -8048581381 / 2820520608, 8663915743 / 2820520608, -12715105075 / 11282082432], [0, 0, 0, 0],
                   [0, 131558114200 / 32700410799, -68118460800 / 10900136933, 87487479700 / 32700410799],
                   [0, -1754552775 / 470086768, 14199869525 / 1410260304, -10690763975 / 1880347072],
                   [0, 127303824393 / 49829197408, -318862633887 / 49829197408, 701980252875 / 199316789632],
                   [0, -282668133 / 205662961, 2019193451 / 616988883, -1453857185 / 822651844],
                   [0, 40617522 / 29380423, -110615467 / 29380423, 69997945 / 29380423]])

    def _segment(self, k):
        """(t_old, h, y_old, Q) of the step that starts at point k: the step's stages formed again on the host from the
        metric's own 8-D equations -- the same dense output solve_ivp keeps per step (RkDenseOutput)."""
        seg = self._segments.get(k)
        if seg is None:
            last = k == len(self.t) - 2
            t_end = self._t_step_end if (last and self.status == 1 and self._t_step_end is not None) else self.t[k + 1]
            t0, h, y0 = self.t[k], t_end - self.t[k], self.y[:, k]
            f = self._metric.geodesic_equations
            K = np.empty((7, 8))
            K[0] = f(t0, y0)
            for s_ in range(1, 6):
                K[s_] = f(t0 + self._C[s_] * h, y0 + h * (np.array(self._A[s_]) @ K[:s_]))
            K[6] = f(t0 + h, y0 + h * (self._B @ K[:6]))
            seg = self._segments[k] = (t0, h, y0, K.T @ self._P)
        return seg

    def sol(self, t):
        """The track as a function of the affine parameter, like `solution.sol` of the reference's
        `solve_ivp(..., dense_output=True)` (geodesic_tracer.py:57-67): -> (8,) for a scalar t, (8, n) for an array, valid
        on [t[0], t[-1]].  Each step's 4th-order dense output is rebuilt on the host from the stored points (7 evaluations
        of metric.geodesic_equations per step touched, cached); nothing in the reference calls it -- it is here so that the
        object a caller gets back has the surface solve_ivp's has."""
        if self._metric is None:
            raise RuntimeError("this Track was built without its metric: no dense output")
        if self.truncated:
            raise RuntimeError("the record was truncated (max_points): steps are missing, no dense output")
        ts = np.atleast_1d(np.asarray(t, dtype=np.float64))
        if len(self.t) < 2:
            out = np.repeat(self.y[:, :1], ts.size, axis=1)
        else:
            seg = np.clip(np.searchsorted(self.t, ts, side="right") - 1, 0, len(self.t) - 2)
            out = np.empty((8, ts.size))
            for j, (k, tj) in enumerate(zip(seg, ts)):
                t0, h, y0, Q = self._segment(int(k))
                x = (tj - t0) / h
                out[:, j] = y0 + h * (Q @ (x ** np.arange(1, 5)))
        return out[:, 0] if np.ndim(t) == 0 else out


def _lt_metric(metric):
    kind = getattr(metric, "is_spherically_symmetric", None)
    if kind is None or not hasattr(metric, "M"):
        raise TypeError("the GPU path integrates Schwarzschild and Kerr metrics only; use integrate_geodesic")
    kerr = not kind
    return ltrace.Metric(ltrace.METRIC_KERR if kerr else ltrace.METRIC_SCHWARZSCHILD, 0, float(metric.M),
                         float(getattr(metric, "a", 0.0)) if kerr else 0.0)


def integrate_geodesics(metric, states0, lambda_max=1000.0, r_stop_inner=None, r_stop_outer=None, max_points=1024,
                        rtol=IVP["rtol"], atol=IVP["atol"], max_step=IVP["max_step"]):
    """`integrate_geodesic` for a batch of 8-D initial states on the GPU -> list of (Track, outcome).
    Defaults as there: stop at the metric's capture radius, at twice each track's start radius, or at
    lambda_max.  The outcome rule is the reference's (final r <= 1.1 r_stop_inner: captured)."""
    ltrace.require_gpu()
    s0 = np.ascontiguousarray(states0, dtype=np.float64).reshape(-1, 8)
    r_in = metric.capture_radius() if r_stop_inner is None else float(r_stop_inner)
    opts = ltrace.default_dense_opts(lambda_max=float(lambda_max), r_stop_inner=r_in,
                                     r_stop_outer=None if r_stop_outer is None else float(r_stop_outer),
                                     rtol=float(rtol), atol=float(atol), max_step=float(max_step),
                                     max_points=int(max_points))
    t, y, count, status, nfev = ltrace.integrate_dense(_lt_metric(metric), s0, opts)
    out = []
    for i in range(s0.shape[0]):
        m = min(int(count[i]), int(max_points))
        step_end = float(t[i, m]) if (status[i] > 0 and count[i] < max_points) else None   # one value past the record: see lt_integrate_dense
        trk = Track(t[i, :m].copy(), np.ascontiguousarray(y[i, :m].T), int(status[i]), nfev[i], count[i] > max_points,
                    metric=metric, t_step_end=step_end)
        out.append((trk, "captured" if trk.y[1, -1] <= 1.1 * r_in else "escaped"))
    return out


def trace_rays(metric, r_obs, alphas, thetas=None, **kwargs):
    """`trace_ray` for many viewing angles at once on the GPU -> list of (Track, outcome) or (None, 'invalid')."""
    alphas = np.atleast_1d(np.asarray(alphas, dtype=np.float64))
    thetas = np.zeros_like(alphas) if thetas is None else np.broadcast_to(np.asarray(thetas, dtype=np.float64), alphas.shape)
    states = [metric.initial_conditions(r_obs, al, th) for al, th in zip(alphas, thetas)]
    live = [i for i, s in enumerate(states) if s is not None]
    res = [(None, "invalid")] * len(states)
    if live:
        for i, r in zip(live, integrate_geodesics(metric, [states[i] for i in live], **kwargs)):
            res[i] = r
    return res


# ---------------------------------------------------------------------------------------------
# drawing helpers (shared with main.py)
# ---------------------------------------------------------------------------------------------
def new_axes(size):
    import matplotlib
    matplotlib.use("Agg", force=False)
    import matplotlib.pyplot as plt
    return plt.subplots(figsize=(size, size))


def draw_scene(ax, metric, r_obs, observer_size=10, observer_label=None):
    ring = np.linspace(0, 2 * np.pi, 200)
    for radius, style, label in ((metric.capture_radius(), None, "Event horizon"),
                                 (getattr(metric, "R_PHOTON", None), "r--", "Photon sphere")):
        if radius is None:
            continue
        if style is None:
            ax.fill(radius * np.cos(ring), radius * np.sin(ring), "k", label=label)
        else:
            ax.plot(radius * np.cos(ring), radius * np.sin(ring), style, linewidth=1.5, label=label)
    ax.plot(r_obs, 0, "go", markersize=observer_size, label=observer_label or f"Observer (r={r_obs}M)")


def draw_track(ax, solution, outcome, label, width=1.2):
    r, phi = solution.y[1], solution.y[3]
    escaped = outcome == "escaped"
    ax.plot(r * np.cos(phi), r * np.sin(phi), color="steelblue" if escaped else "crimson",
            linestyle="-" if escaped or width > 1.5 else "--", linewidth=width, label=label)


def finish_axes(ax, title, half_width=None, big=False):
    ax.set_title(title, fontsize=14 if big else None)
    ax.set_xlabel("x / M", fontsize=12 if big else None)
    ax.set_ylabel("y / M", fontsize=12 if big else None)
    if half_width is not None:
        ax.set_xlim(-half_width, half_width)
        ax.set_ylim(-half_width, half_width)
    ax.set_aspect("equal")
    ax.legend(loc="upper left", fontsize=10 if big else 8)
    ax.grid(True, alpha=0.3)


def save(fig, path, dpi=150, tight=False):
    import matplotlib.pyplot as plt
    plt.tight_layout()
    fig.savefig(path, dpi=dpi, **({"bbox_inches": "tight"} if tight else {}))
    plt.close(fig)


def plot_trajectories(metric, r_obs, angles_deg, ax=None, backend="scipy"):
    """Fan of equatorial tracks (x = r cos phi, y = r sin phi), one per viewing angle in degrees.
    backend "scipy": one solve_ivp per ray like the reference; "hip": the whole fan in one GPU launch."""
    if backend not in ("scipy", "hip"):
        raise ValueError("backend must be 'scipy' or 'hip'")
    if ax is None:
        _, ax = new_axes(10)
    draw_scene(ax, metric, r_obs)
    fan = trace_rays(metric, r_obs, np.radians(angles_deg)) if backend == "hip" else None
    for j, deg in enumerate(angles_deg):
        sol, outcome = fan[j] if fan is not None else trace_ray(metric, r_obs, np.radians(deg))
        if sol is not None:
            draw_track(ax, sol, outcome, f"α={deg}° ({outcome})")
    finish_axes(ax, f"Photon trajectories (critical angle ≈ {np.degrees(metric.alpha_crit(r_obs)):.2f}°)")
    return ax


def demo(metric=None, r_obs=None, angles=DEMO_ANGLES, output="geodesic_trajectories.png", backend="scipy"):
    metric = Schwarzschild(M=1.0) if metric is None else metric
    r_obs = 50.0 * metric.M if r_obs is None else r_obs
    bar = "=" * 60
    print(f"{bar}\nGeodesic Tracer\n{bar}\nMetric: {type(metric).__name__}\nObserver radius: r_obs = {r_obs} M")
    print(f"Critical viewing angle: {np.degrees(metric.alpha_crit(r_obs)):.4f}°\n{bar}\n\nTracing rays:\n" + "-" * 40)
    rows = []
    fan = trace_rays(metric, r_obs, np.radians(angles)) if backend == "hip" else None
    for j, deg in enumerate(angles):
        b = metric.viewing_angle_to_impact_parameter(np.radians(deg), r_obs)
        _, outcome = fan[j] if fan is not None else trace_ray(metric, r_obs, np.radians(deg))
        rows.append((deg, float(b), outcome))
        print(f"  α = {deg:6.2f}°  →  b = {b:6.3f} M  →  {'CAPTURED' if outcome == 'captured' else 'ESCAPED'}")
    if output:
        print("\nGenerating plot...")
        fig, ax = new_axes(12)
        plot_trajectories(metric, r_obs, angles, ax=ax, backend=backend)
        ax.set_xlim(-0.3 * r_obs, 1.2 * r_obs)
        ax.set_ylim(-0.5 * r_obs, 0.5 * r_obs)
        save(fig, output, dpi=150, tight=True)
        print(f"Saved: {output}")
    return rows


if __name__ == "__main__":
    import argparse
    ap = argparse.ArgumentParser(description="fan of photon trajectories around a Schwarzschild black hole")
    ap.add_argument("--backend", choices=("scipy", "hip"), default="scipy")
    ap.add_argument("--output", default="geodesic_trajectories.png")
    args = ap.parse_args()
    demo(backend=args.backend, output=args.output)
