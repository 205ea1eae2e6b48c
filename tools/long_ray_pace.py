#!/usr/bin/env python3
"""Pace of the serial chain that bounds small launches: find the longest rays of the benchmark frame, then trace
each ALONE on the chip (one wavefront: the ray replicated 64 times through lt_trace_batch_kerr) and report
microseconds per RK4 step.  usage: long_ray_pace.py [size]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "light-path-tracer_amd"))
import ltrace
size = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
fov = np.radians(40.0)
cam = ltrace.Camera(size, size, fov, fov, 0.0, 0.0, 50.0, np.pi / 2)
met = ltrace.Metric(1, 0, 1.0, 0.9)
out = ltrace.render(cam, met, ltrace.default_opts(precision=32), want=("steps",))
steps = out["steps"]
alpha, theta, cols = ltrace.pixel_angles(cam)
order = np.argsort(steps.ravel())[::-1][:6]
def run(al, th, refine, n=64):
    fa = np.empty(n); w = np.empty(n, dtype=np.int64); ev = np.empty(n, dtype=np.uint32)
    a = np.full(n, al, dtype=np.float64); t = np.full(n, th); r = np.full(n, refine, dtype=np.uint8)
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        ltrace.trace_batch_kerr(1.0, 0.9, 50.0, a, t, np.pi / 2, 5000.0, r, fa, w, integrator="rk4", precision=32, out_rhs_evals=ev)
        best = min(best, time.perf_counter() - t0)
    return best, int(ev[0]) // 4
base, _ = run(0.3, 1.0, 0)   # a short ray: the call's fixed cost
print(f"fixed cost of the call (short ray): {base * 1e3:.3f} ms")
for p in order:
    iy, ix = divmod(int(p), size)
    dt, n = run(float(alpha[iy, ix]), float(theta[iy, ix]), int(cols[ix]))
    print(f"pixel ({iy},{ix}): {steps[iy, ix]} steps in the frame, {n} alone; {(dt - base) * 1e3:.3f} ms alone -> {(dt - base) * 1e6 / max(n, 1):.3f} us per step")
