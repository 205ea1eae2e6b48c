"""One photon around a black hole: the reference's `main.py` entry point (a single ray at
alpha = 8 degrees from r_obs = 50 M, dense solve_ivp integration, printed summary, x-y plot).

Backend additions: `--alpha`, `--a` (Kerr spin) and `--output` on the command line; the drawing
itself lives in geodesic_tracer (shared with its trajectory fan)."""
import argparse

import numpy as np

import geodesic_tracer as gt
from metrics import Kerr, Schwarzschild

SUMMARY = ("Metric:             {name}\n"
           "Observer radius:    r_obs = {r_obs} M\n"
           "Viewing angle:      α = {deg}°\n"
           "Impact parameter:   b = {b:.4f} M\n"
           "Outcome:            {outcome}")


def main(metric=None, alpha_deg=8.0, output="example_geodesic.png"):
    metric = Schwarzschild(M=1.0) if metric is None else metric
    r_obs, alpha = 50.0 * metric.M, np.radians(alpha_deg)
    solution, outcome = gt.trace_ray(metric, r_obs, alpha)
    b = metric.viewing_angle_to_impact_parameter(alpha, r_obs)
    print(SUMMARY.format(name=type(metric).__name__, r_obs=r_obs, deg=alpha_deg, b=b, outcome=outcome.upper()))
    if solution is not None and output:
        fig, ax = gt.new_axes(10)
        gt.draw_scene(ax, metric, r_obs, observer_size=12, observer_label="Observer")
        gt.draw_track(ax, solution, outcome, f"Photon path ({outcome})", width=2)
        gt.finish_axes(ax, f"{type(metric).__name__} geodesic (α = {alpha_deg}°, b = {b:.2f} M)",
                       half_width=r_obs * 1.1, big=True)
        gt.save(fig, output, dpi=150)
        print(f"\nSaved: {output}")
    return solution, outcome


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--alpha", type=float, default=8.0, help="viewing angle in degrees")
    ap.add_argument("--a", type=float, default=0.0, help="spin (0 = Schwarzschild)")
    ap.add_argument("--output", default="example_geodesic.png")
    args = ap.parse_args()
    main(Kerr(1.0, args.a, integrator="rk4") if args.a else None, args.alpha, args.output)
