"""Dense single-ray integration and trajectory plots: the reference's geodesic_tracer.py surface
(`integrate_geodesic`, `trace_ray`, `plot_trajectories`, the demo under __main__).

This is the trajectory side of the metric plugin -- one to ten rays through scipy.solve_ivp on the
metric's 8-D `geodesic_equations`, for plotting -- not the per-pixel path; it stays on the host
(SURVEY 8f-3 ranks a batched GPU version as future work).  Units: G = c = 1.
"""
import numpy as np
from scipy.integrate import solve_ivp

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


def plot_trajectories(metric, r_obs, angles_deg, ax=None):
    """Fan of equatorial tracks (x = r cos phi, y = r sin phi), one per viewing angle in degrees."""
    if ax is None:
        _, ax = new_axes(10)
    draw_scene(ax, metric, r_obs)
    for deg in angles_deg:
        sol, outcome = trace_ray(metric, r_obs, np.radians(deg))
        if sol is not None:
            draw_track(ax, sol, outcome, f"α={deg}° ({outcome})")
    finish_axes(ax, f"Photon trajectories (critical angle ≈ {np.degrees(metric.alpha_crit(r_obs)):.2f}°)")
    return ax


def demo(metric=None, r_obs=None, angles=DEMO_ANGLES, output="geodesic_trajectories.png"):
    metric = Schwarzschild(M=1.0) if metric is None else metric
    r_obs = 50.0 * metric.M if r_obs is None else r_obs
    bar = "=" * 60
    print(f"{bar}\nGeodesic Tracer\n{bar}\nMetric: {type(metric).__name__}\nObserver radius: r_obs = {r_obs} M")
    print(f"Critical viewing angle: {np.degrees(metric.alpha_crit(r_obs)):.4f}°\n{bar}\n\nTracing rays:\n" + "-" * 40)
    rows = []
    for deg in angles:
        b = metric.viewing_angle_to_impact_parameter(np.radians(deg), r_obs)
        _, outcome = trace_ray(metric, r_obs, np.radians(deg))
        rows.append((deg, float(b), outcome))
        print(f"  α = {deg:6.2f}°  →  b = {b:6.3f} M  →  {'CAPTURED' if outcome == 'captured' else 'ESCAPED'}")
    if output:
        print("\nGenerating plot...")
        fig, ax = new_axes(12)
        plot_trajectories(metric, r_obs, angles, ax=ax)
        ax.set_xlim(-0.3 * r_obs, 1.2 * r_obs)
        ax.set_ylim(-0.5 * r_obs, 0.5 * r_obs)
        save(fig, output, dpi=150, tight=True)
        print(f"Saved: {output}")
    return rows


if __name__ == "__main__":
    demo()
