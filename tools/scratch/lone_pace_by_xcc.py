#!/usr/bin/env python3
"""Pace of one lone wavefront (the 7484-step ray replicated x64) over repeated launches, with the XCD it ran on
(wave stamps).  LT_STAMPS_FILE must be set."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "light-path-tracer_amd"))
import ltrace
size = 4096
fov = np.radians(40.0)
cam = ltrace.Camera(size, size, fov, fov, 0.0, 0.0, 50.0, np.pi / 2)
alpha, theta, cols = ltrace.pixel_angles(cam)
py, px = 402, 2046
al = np.full(64, float(alpha[py, px])); th = np.full(64, float(theta[py, px])); rf = np.full(64, int(cols[px]), np.uint8)
if len(sys.argv) > 2 and sys.argv[2] == "tile":          # the real 8x8 tile instead of 64 copies of its longest ray
    y0, x0 = py // 8 * 8, px // 8 * 8
    al = alpha[y0:y0 + 8, x0:x0 + 8].astype(np.float64).ravel(); th = theta[y0:y0 + 8, x0:x0 + 8].ravel()
    rf = np.repeat(cols[x0:x0 + 8][None, :], 8, 0).astype(np.uint8).ravel()
if len(sys.argv) > 2 and sys.argv[2] == "one":           # the longest ray in lane 0, 63 short rays beside it
    al = np.full(64, 0.3); th = np.full(64, 1.0); rf = np.zeros(64, np.uint8)
    al[0], th[0], rf[0] = float(alpha[py, px]), float(theta[py, px]), int(cols[px])
f = os.environ["LT_STAMPS_FILE"]
gap = float(sys.argv[1]) if len(sys.argv) > 1 else 0.0
rows = []
for rep in range(24):
    fa, w = np.empty(64), np.empty(64, dtype=np.int64)
    ltrace.trace_batch_kerr(1.0, 0.9, 50.0, al, th, np.pi / 2, 5000.0, rf, fa, w, integrator="rk4", precision=32)
    a = np.fromfile(f, dtype=np.uint32).reshape(-1, 4)
    rows.append((int(a[0, 3] & 0xf), a[0, 1] / 100 / (a[0, 3] >> 4), a[0, 2] / max(a[0, 1], 1) * 100))
    if gap:
        time.sleep(gap)
print("xcc us/step s_memtime-MHz:", " ".join(f"{x}:{p:.3f}:{c:.0f}" for x, p, c in rows))
by = {}
for x, p, c in rows:
    by.setdefault(x, []).append(p)
print("by XCD:", {x: (round(min(v), 3), round(max(v), 3), len(v)) for x, v in sorted(by.items())})
