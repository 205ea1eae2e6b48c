"""One photon around a black hole -- same entry point as the reference's main.py: trace a single ray
at alpha = 8 degrees from r_obs = 50 M with the dense solve_ivp integrator, print its impact
parameter and outcome, plot the track."""
import numpy as np

from geodesic_tracer import trace_ray
from metrics import Schwarzschild


def main(metric=None, alpha_deg=8.0, output="example_geodesic.png"):
    import matplotlib
    matplotlib.use("Agg", force=False)
    import matplotlib.pyplot as plt

    if metric is None:
        metric = Schwarzschild(M=1.0)
    r_obs = 50.0 * metric.M
    alpha = np.radians(alpha_deg)
    solution, outcome = trace_ray(metric, r_obs, alpha)
    b = metric.viewing_angle_to_impact_parameter(alpha, r_obs)
    print(f"Metric:             {type(metric).__name__}")
    print(f"Observer radius:    r_obs = {r_obs} M")
    print(f"Viewing angle:      α = {alpha_deg}°")
    print(f"Impact parameter:   b = {b:.4f} M")
    print(f"Outcome:            {outcome.upper()}")

    r, phi = solution.y[1], solution.y[3]
    fig, ax = plt.subplots(figsize=(10, 10))
    t = np.linspace(0, 2 * np.pi, 200)
    rh = metric.capture_radius()
    ax.fill(rh * np.cos(t), rh * np.sin(t), "k", label="Event horizon")
    if hasattr(metric, "R_PHOTON"):
        ax.plot(metric.R_PHOTON * np.cos(t), metric.R_PHOTON * np.sin(t), "r--", linewidth=1.5, label="Photon sphere")
    ax.plot(r * np.cos(phi), r * np.sin(phi), color="steelblue" if outcome == "escaped" else "crimson",
            linewidth=2, label=f"Photon path ({outcome})")
    ax.plot(r_obs, 0, "go", markersize=12, label="Observer")
    ax.set_xlabel("x / M", fontsize=12)
    ax.set_ylabel("y / M", fontsize=12)
    ax.set_title(f"{type(metric).__name__} geodesic (α = {alpha_deg}°, b = {b:.2f} M)", fontsize=14)
    lim = r_obs * 1.1
    ax.set_xlim(-lim, lim)
    ax.set_ylim(-lim, lim)
    ax.set_aspect("equal")
    ax.legend(loc="upper left", fontsize=10)
    ax.grid(True, alpha=0.3)
    plt.tight_layout()
    plt.savefig(output, dpi=150)
    print(f"\nSaved: {output}")
    return solution, outcome


if __name__ == "__main__":
    main()
