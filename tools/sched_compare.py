#!/usr/bin/env python3
"""direct vs queue schedule of the integrate kernel across frame sizes and integrators."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "light-path-tracer_amd"))
import ltrace
fov = np.radians(40.0)
met = ltrace.Metric(1, 0, 1.0, 0.9)
print(f"{'frame':>6s} {'integrator':>10s} {'direct ms':>10s} {'queue ms':>10s} {'lane util (steps/64/maxsteps per tile)':>40s}")
for integ, prec in (("rk4", 32), ("dp45", 64)):
    for n in (128, 256, 512, 1024, 2048):
        cam = ltrace.Camera(n, n, fov, fov, 0.0, 0.0, 50.0, np.pi / 2)
        ms = {}
        for sched in ("direct", "queue"):
            o = ltrace.default_opts(integrator=integ, precision=prec, schedule=sched)
            best = 1e9
            for _ in range(4):
                out = ltrace.render(cam, met, o, want=("steps",))
                best = min(best, out["stats"]["integrate_ms"])
            ms[sched] = best
        s = out["steps"].astype(np.int64)
        t = s.reshape(n // 8, 8, n // 8, 8).max(axis=(1, 3))
        print(f"{n:6d} {integ:>10s} {ms['direct']:10.3f} {ms['queue']:10.3f} {s.sum() / (64 * t.sum()):40.3f}")
