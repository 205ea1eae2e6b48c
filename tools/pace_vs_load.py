#!/usr/bin/env python3
"""Pace of one long ray (RK4 steps per microsecond) as a function of how busy the chip is: the same 7 873-step ray
replicated into 1 ... 5120 identical wavefronts through lt_trace_batch_kerr.  Shows what a long ray loses while the
bulk of a frame is still resident (clock drop under load, then SIMD sharing)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "light-path-tracer_amd"))
import ltrace
size = 4096
fov = np.radians(40.0)
cam = ltrace.Camera(size, size, fov, fov, 0.0, 0.0, 50.0, np.pi / 2)
alpha, theta, cols = ltrace.pixel_angles(cam)
iy, ix = 814, 2053     # 7873 steps
al, th, rf = float(alpha[iy, ix]), float(theta[iy, ix]), int(cols[ix])
def run(nw):
    n = 64 * nw
    fa = np.empty(n); w = np.empty(n, dtype=np.int64); ev = np.empty(n, dtype=np.uint32)
    a = np.full(n, al); t = np.full(n, th); r = np.full(n, rf, dtype=np.uint8)
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        ltrace.trace_batch_kerr(1.0, 0.9, 50.0, a, t, np.pi / 2, 5000.0, r, fa, w, integrator="rk4", precision=32, out_rhs_evals=ev)
        best = min(best, time.perf_counter() - t0)
    return best, int(ev[0]) // 4
for nw in (1, 64, 256, 512, 1024, 2048, 4096, 5120):
    dt, steps = run(nw)
    print(f"{nw:5d} identical waves ({nw / 1024:.2f} per SIMD, {nw / 256:.1f} per CU): {dt * 1e3:8.3f} ms for {steps} steps -> {dt * 1e6 / steps:.3f} us per step")
