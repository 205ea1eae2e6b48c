#!/usr/bin/env python3
"""Project multi-GPU strong scaling on ONE GPU: render each row partition of the benchmark frame in
turn and report the integrate-kernel time of every partition (the N-GPU time is the max over parts,
plus the gather)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "light-path-tracer_amd"))
import ltrace
size = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
integ = sys.argv[2] if len(sys.argv) > 2 else "rk4"   # rk4 (float32) or dp45 (float64)
fov = np.radians(40.0)
cam = ltrace.Camera(size, size, fov, fov, 0.0, 0.0, 50.0, np.pi / 2)
met = ltrace.Metric(1, 0, 1.0, 0.9)
for n in (1, 2, 4, 8):
    ts, mx = [], []
    for p in range(n):
        o = ltrace.default_opts(integrator=integ, precision=32 if integ == "rk4" else 64, n_parts=n, part=p, row_block=16)
        ltrace.render(cam, met, o, want=("status",))
        out = ltrace.render(cam, met, o, want=("status", "steps"))
        ts.append(out["stats"]["integrate_ms"]); mx.append(int(out["steps"].max()))
    print(f"n_parts={n}: integrate ms per part = {[round(t, 2) for t in ts]}  max steps per part = {mx}  -> max {max(ts):.2f} ms, "
          f"speed-up of the integrate kernel {ts[0] if n == 1 else 0:.0f}" if n == 1 else
          f"n_parts={n}: integrate ms per part = {[round(t, 2) for t in ts]}  max steps per part = {mx}  -> max {max(ts):.2f} ms")
