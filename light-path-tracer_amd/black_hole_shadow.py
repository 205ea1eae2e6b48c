"""Black-hole shadow image -- same entry point as the reference's black_hole_shadow.py.

`main()` reproduces the reference: an 800x800 analytic Schwarzschild shadow (a pixel is black when
its viewing angle is below alpha_crit), saved as black_hole_shadow.png.  No ray is integrated there,
so it is a few vectorised numpy lines on the host.

`render_traced()` / `main(traced=True)` is the backend's addition: the shadow obtained by actually
tracing one ray per pixel on the GPU (needed for Kerr, whose analytic alpha_crit is only a
conservative circle -- reference metrics.py:893-896).  It uses the pinhole camera of image_lens.py,
not the cos(ax)cos(ay) mapping of the analytic mode (the reference's two scripts differ there).
"""
import numpy as np

import ltrace
from metrics import Schwarzschild


def pixel_to_viewing_angle(i, n, fov):
    """Angle of pixel index i (scalar or array) along one axis of an n-pixel, fov-wide image."""
    return np.arctan((i - n / 2) / (n / 2) * np.tan(fov / 2))


def get_pixel_color(metric, r_obs, alpha, alpha_crit):
    return 0.0 if alpha < alpha_crit else 1.0


def analytic_shadow(metric, width=800, height=800, fov_deg=40, r_obs=None):
    """(width, height) float64 array indexed [x, y] like the reference's `image[i, j]`."""
    fov = np.radians(fov_deg)
    r_obs = 50.0 * metric.M if r_obs is None else r_obs
    alpha_crit = metric.alpha_crit(r_obs)
    ax = pixel_to_viewing_angle(np.arange(width), width, fov)
    ay = pixel_to_viewing_angle(np.arange(height), height, fov)
    alpha = np.arccos(np.cos(ax)[:, None] * np.cos(ay)[None, :])
    return np.where(alpha < alpha_crit, 0.0, 1.0)


def render_traced(metric, width=800, height=800, fov_deg=40, r_obs=None, psi=(0.0, 0.0), integrator=None,
                  precision=None, schedule=None):
    """(height, width) float32 image, 0 where the traced ray did not escape, 1 where it did; plus the
    per-pixel status and the render statistics."""
    vfov = np.radians(fov_deg)
    hfov = 2 * np.arctan(np.tan(vfov / 2) * width / height)
    r_obs = 50.0 * metric.M if r_obs is None else r_obs
    kerr = not metric.is_spherically_symmetric
    cam = ltrace.Camera(width, height, hfov, vfov, psi[0], psi[1], r_obs, np.pi / 2)
    met = ltrace.Metric(ltrace.METRIC_KERR if kerr else ltrace.METRIC_SCHWARZSCHILD, 0, float(metric.M),
                        float(getattr(metric, "a", 0.0)))
    opts = ltrace.default_opts(integrator=integrator or getattr(metric, "integrator", "rk4"),
                               precision=precision or getattr(metric, "precision", 32),
                               schedule=schedule or getattr(metric, "schedule", "direct"))
    out = ltrace.render(cam, met, opts, want=("status",))
    return (out["status"] == 1).astype(np.float32), out["status"], out["stats"]


def main(metric=None, traced=False, width=800, height=800, output="black_hole_shadow.png"):
    import matplotlib
    matplotlib.use("Agg", force=False)
    import matplotlib.pyplot as plt

    if metric is None:
        metric = Schwarzschild(M=1.0)
    if traced:
        image, _, stats = render_traced(metric, width, height)
        print(f"traced {stats['rays']:,} rays: {stats['captured']:,} captured, {stats['invalid']:,} invalid, "
              f"integrate kernel {stats['integrate_ms']:.3f} ms")
        plt.imshow(image, cmap="gray", origin="upper")
    else:
        plt.imshow(analytic_shadow(metric, width, height), cmap="gray", origin="lower")
    plt.axis("off")
    plt.savefig(output, dpi=200, bbox_inches="tight")
    plt.close()


if __name__ == "__main__":
    import argparse
    from metrics import Kerr
    ap = argparse.ArgumentParser()
    ap.add_argument("--traced", action="store_true", help="trace one ray per pixel on the GPU")
    ap.add_argument("--a", type=float, default=0.0, help="spin for the traced mode (0 = Schwarzschild)")
    ap.add_argument("--size", type=int, default=800)
    args = ap.parse_args()
    main(Kerr(1.0, args.a) if args.a else None, traced=args.traced or args.a != 0, width=args.size, height=args.size)
