"""Single-ray geodesic integration and trajectory plots -- same surface as the reference's
geodesic_tracer.py (integrate_geodesic, trace_ray, plot_trajectories).

This is the dense-trajectory side of the metric plugin: one to ten rays, integrated with
scipy.solve_ivp on the metric's 8-D `geodesic_equations`, for plotting.  It is not the per-pixel
path and stays on the host (SURVEY 8f ranks a batched GPU version as future work).

Units: G = c = 1.
"""
import numpy as np
from scipy.integrate import solve_ivp

from metrics import Schwarzschild


def integrate_geodesic(metric, state0, lambda_max=1000.0, r_stop_inner=None, r_stop_outer=None):
    """Integrate from the 8-D state0 until capture (r <= r_stop_inner, default metric.capture_radius()),
    escape (r >= r_stop_outer, default 2 r_0) or lambda_max.  -> (OdeSolution, 'captured' | 'escaped')."""
    r_in = metric.capture_radius() if r_stop_inner is None else r_stop_inner
    r_out = state0[1] * 2.0 if r_stop_outer is None else r_stop_outer

    def hit_inner(_lam, y):
        return y[1] - r_in
    hit_inner.terminal, hit_inner.direction = True, -1

    def hit_outer(_lam, y):
        return y[1] - r_out
    hit_outer.terminal, hit_outer.direction = True, 1

    sol = solve_ivp(metric.geodesic_equations, [0, lambda_max], state0, method="RK45",
                    events=[hit_inner, hit_outer], max_step=1.0, rtol=1e-8, atol=1e-10, dense_output=True)
    return sol, ("captured" if sol.y[1, -1] <= r_in * 1.1 else "escaped")


def trace_ray(metric, r_obs, alpha, **kwargs):
    """One ray from its viewing angle -> (solution, outcome) or (None, 'invalid')."""
    state0 = metric.initial_conditions(r_obs, alpha)
    if state0 is None:
        return None, "invalid"
    return integrate_geodesic(metric, state0, **kwargs)


def _draw_hole(ax, metric):
    t = np.linspace(0, 2 * np.pi, 200)
    rh = metric.capture_radius()
    ax.fill(rh * np.cos(t), rh * np.sin(t), "k", label="Event horizon")
    if hasattr(metric, "R_PHOTON"):
        ax.plot(metric.R_PHOTON * np.cos(t), metric.R_PHOTON * np.sin(t), "r--", linewidth=1.5, label="Photon sphere")


def plot_trajectories(metric, r_obs, angles_deg, ax=None):
    """Equatorial-plane tracks (x = r cos phi, y = r sin phi) for a list of viewing angles in degrees."""
    import matplotlib.pyplot as plt
    if ax is None:
        _, ax = plt.subplots(figsize=(10, 10))
    _draw_hole(ax, metric)
    ax.plot(r_obs, 0, "go", markersize=10, label=f"Observer (r={r_obs}M)")
    for deg in angles_deg:
        sol, outcome = trace_ray(metric, r_obs, np.radians(deg))
        if sol is None:
            continue
        r, phi = sol.y[1], sol.y[3]
        esc = outcome == "escaped"
        ax.plot(r * np.cos(phi), r * np.sin(phi), color="steelblue" if esc else "crimson",
                linestyle="-" if esc else "--", linewidth=1.2, label=f"α={deg}° ({outcome})")
    ax.set_title(f"Photon trajectories (critical angle ≈ {np.degrees(metric.alpha_crit(r_obs)):.2f}°)")
    ax.set_xlabel("x / M")
    ax.set_ylabel("y / M")
    ax.set_aspect("equal")
    ax.legend(loc="upper left", fontsize=8)
    ax.grid(True, alpha=0.3)
    return ax


if __name__ == "__main__":
    import matplotlib
    matplotlib.use("Agg", force=False)
    import matplotlib.pyplot as plt

    metric = Schwarzschild(M=1.0)
    r_obs = 50.0 * metric.M
    angles = [0, 2, 4, 5, 5.5, 5.97, 6.5, 8, 10, 15]
    print("=" * 60 + "\nGeodesic Tracer\n" + "=" * 60)
    print(f"Metric: {type(metric).__name__}\nObserver radius: r_obs = {r_obs} M")
    print(f"Critical viewing angle: {np.degrees(metric.alpha_crit(r_obs)):.4f}°\n" + "=" * 60)
    print("\nTracing rays:\n" + "-" * 40)
    for deg in angles:
        b = metric.viewing_angle_to_impact_parameter(np.radians(deg), r_obs)
        _, outcome = trace_ray(metric, r_obs, np.radians(deg))
        print(f"  α = {deg:6.2f}°  →  b = {b:6.3f} M  →  {'CAPTURED' if outcome == 'captured' else 'ESCAPED'}")
    fig, ax = plt.subplots(figsize=(12, 10))
    plot_trajectories(metric, r_obs, angles, ax=ax)
    ax.set_xlim(-r_obs * 0.3, r_obs * 1.2)
    ax.set_ylim(-r_obs * 0.5, r_obs * 0.5)
    plt.tight_layout()
    plt.savefig("geodesic_trajectories.png", dpi=150, bbox_inches="tight")
    print("Saved: geodesic_trajectories.png")
