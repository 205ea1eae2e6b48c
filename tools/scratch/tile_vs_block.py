#!/usr/bin/env python3
"""The same 8x8 tile, traced (a) through lt_trace_batch_kerr as one wavefront, (b) inside its 16-row block through
lt_render (n_parts = 256): wave stamps of the longest wave.  LT_STAMPS_FILE must be set (the library dumps after each launch)."""
import os, sys, subprocess
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "light-path-tracer_amd"))
import ltrace
size = 4096
fov = np.radians(40.0)
cam = ltrace.Camera(size, size, fov, fov, 0.0, 0.0, 50.0, np.pi / 2)
alpha, theta, cols = ltrace.pixel_angles(cam)
py, px = 402, 2046
y0, x0 = py // 8 * 8, px // 8 * 8
al = alpha[y0:y0 + 8, x0:x0 + 8].astype(np.float64).ravel(); th = theta[y0:y0 + 8, x0:x0 + 8].ravel()
rf = np.repeat(cols[x0:x0 + 8][None, :], 8, 0).astype(np.uint8).ravel()
f = os.environ["LT_STAMPS_FILE"]
def show(tag):
    a = np.fromfile(f, dtype=np.uint32).reshape(-1, 4)
    i = int(np.argmax(a[:, 1]))
    dur, cyc, steps = int(a[i, 1]), int(a[i, 2]), int(a[i, 3] >> 4)
    print(f"{tag}: waves {len(a)}, longest wave {i}: {dur / 100:.1f} us, steps {steps}, {dur / 100 / steps:.4f} us/step, {cyc / steps:.0f} cycles/step, clock {cyc / dur * 100:.0f} MHz")
for rep in range(3):
    fa, w = np.empty(64), np.empty(64, dtype=np.int64)
    ltrace.trace_batch_kerr(1.0, 0.9, 50.0, al, th, np.pi / 2, 5000.0, rf, fa, w, integrator="rk4", precision=32)
    show("batch, one wave  ")
    out = ltrace.render(cam, ltrace.Metric(1, 0, 1.0, 0.9), ltrace.default_opts(precision=32, n_parts=256, part=25), want=("steps",))
    show("block 25 of 256   ")
