#!/usr/bin/env python3
"""Pace of a lone wavefront as a function of how many of its lanes are still enabled: the longest ray of the 4096^2
frame in k lanes (first k, or spread), short rays in the others; 8 consecutive launches each (consecutive launches of
one workgroup land on 4 CUs in turn).  Needs LT_STAMPS_FILE.  With LT_D_LONG=2000000000 no wave ever switches to ghost
lanes (k_kerr_direct): that is the hardware's behaviour; the default shows what the ghost lanes make of it."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "light-path-tracer_amd"))
import ltrace
size = 4096
fov = np.radians(40.0)
cam = ltrace.Camera(size, size, fov, fov, 0.0, 0.0, 50.0, np.pi / 2)
alpha, theta, cols = ltrace.pixel_angles(cam)
py, px = 402, 2046
A, T, R = float(alpha[py, px]), float(theta[py, px]), int(cols[px])
f = os.environ["LT_STAMPS_FILE"]
def pace(lanes):
    al = np.full(64, 0.3); th = np.full(64, 1.0); rf = np.zeros(64, np.uint8)
    al[lanes], th[lanes], rf[lanes] = A, T, R
    out = []
    for rep in range(8):
        fa, w = np.empty(64), np.empty(64, dtype=np.int64)
        ltrace.trace_batch_kerr(1.0, 0.9, 50.0, al, th, np.pi / 2, 5000.0, rf, fa, w, integrator="rk4", precision=32)
        a = np.fromfile(f, dtype=np.uint32).reshape(-1, 4)
        out.append(a[0, 1] / 100 / (a[0, 3] >> 4))
    return out
for name, lanes in (("1 lane (0)", [0]), ("1 lane (63)", [63]), ("2 lanes (0,1)", [0, 1]), ("2 lanes (0,32)", [0, 32]), ("4 lanes (0-3)", list(range(4))),
                    ("8 lanes (0-7)", list(range(8))), ("16 lanes", list(range(16))), ("32 lanes (0-31)", list(range(32))),
                    ("32 lanes (even)", list(range(0, 64, 2))), ("48 lanes", list(range(48))), ("63 lanes", list(range(63))), ("64 lanes", list(range(64)))):
    p = pace(lanes)
    print(f"{name:18s} us/step over 8 launches: " + " ".join(f"{x:.3f}" for x in p) + f"   min {min(p):.3f} max {max(p):.3f}")
