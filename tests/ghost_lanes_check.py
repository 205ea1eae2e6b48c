"""Prints a digest of every output of a small Kerr frame (float32 and float64 fixed-step RK4, both schedules) and of a
ragged trace_batch call.  tests/test_gpu_ghost_lanes.py runs it with LT_D_LONG=1 / LT_Q_LONG=0 (every wavefront switches
to ghost lanes at once: finished lanes shadow a running one, k_kerr_direct / k_kerr_queue) and with thresholds so large
that no wave ever does; the digests must be equal -- ghost lanes change which lanes are enabled, never a result."""
import hashlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "light-path-tracer_amd"))

import numpy as np   # noqa: E402
import ltrace        # noqa: E402


def main():
    h = hashlib.sha256()
    W, H = 203, 117                      # ragged: last tile column / row partly padding
    fov_v = np.radians(35.0)
    cam = ltrace.Camera(W, H, 2 * np.arctan(np.tan(fov_v / 2) * W / H), fov_v, 0.01, -0.02, 30.0, 1.2)
    for prec in (32, 64):
        for a in (0.0, 0.9, 0.998):
            for sched in ("direct", "queue"):
                o = ltrace.default_opts(precision=prec, schedule=sched)
                out = ltrace.render(cam, ltrace.Metric(1, 0, 1.0, a), o, want=("fa", "winding", "status", "steps", "rgba"))
                for name in ("fa", "winding", "status", "steps", "rgba"):
                    h.update(np.ascontiguousarray(out[name]).tobytes())
    rng = np.random.default_rng(5)
    n = 1000                             # not a multiple of 64
    al, th = rng.uniform(-0.3, 0.3, n), rng.uniform(0.0, 2 * np.pi, n)
    rf = (rng.uniform(size=n) < 0.1).astype(np.uint8)
    for prec in (32, 64):
        fa, w = np.empty(n), np.empty(n, dtype=np.int64)
        for sched in ("direct", "queue"):
            ltrace.trace_batch_kerr(1.0, 0.9, 50.0, al, th, np.pi / 2, 5000.0, rf, fa, w, integrator="rk4", precision=prec,
                                    schedule=sched)
            h.update(fa.tobytes()); h.update(w.tobytes())
    if os.environ.get("LT_CHECK_BIG"):
        # more tiles than the chip has wavefront slots (5 120 for float32 RK4, 2 048 for the float64 integrators): the
        # launches that hand tiles out from a queue head (LT_D_PERSIST, k_kerr_direct)
        W, H = 1100, 900
        cam = ltrace.Camera(W, H, 2 * np.arctan(np.tan(fov_v / 2) * W / H), fov_v, 0.0, 0.0, 50.0, np.pi / 2)
        for integ, prec in (("rk4", 32), ("dp45", 64)):
            out = ltrace.render(cam, ltrace.Metric(1, 0, 1.0, 0.9), ltrace.default_opts(integrator=integ, precision=prec),
                                want=("fa", "winding", "status", "steps", "rgba"))
            for name in ("fa", "winding", "status", "steps", "rgba"):
                h.update(np.ascontiguousarray(out[name]).tobytes())
    print("digest", h.hexdigest(), flush=True)
    return 0


if __name__ == "__main__":
    sys.exit(main())
