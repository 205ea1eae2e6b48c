#!/usr/bin/env python3
"""lone_pace_by_lanes.py with the wave's HW_ID (scratch build that puts HW_REG_HW_ID in stamp word 0; LTRACE_LIB)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "light-path-tracer_amd"))
import ltrace
size = 4096
fov = np.radians(40.0)
cam = ltrace.Camera(size, size, fov, fov, 0.0, 0.0, 50.0, np.pi / 2)
alpha, theta, cols = ltrace.pixel_angles(cam)
py, px = 402, 2046
A, T, R = float(alpha[py, px]), float(theta[py, px]), int(cols[px])
f = os.environ["LT_STAMPS_FILE"]
def pace(lanes, reps=16):
    al = np.full(64, 0.3); th = np.full(64, 1.0); rf = np.zeros(64, np.uint8)
    al[lanes], th[lanes], rf[lanes] = A, T, R
    out = []
    for rep in range(reps):
        fa, w = np.empty(64), np.empty(64, dtype=np.int64)
        ltrace.trace_batch_kerr(1.0, 0.9, 50.0, al, th, np.pi / 2, 5000.0, rf, fa, w, integrator="rk4", precision=32)
        a = np.fromfile(f, dtype=np.uint32).reshape(-1, 4)
        hw = int(a[0, 0])
        out.append((a[0, 1] / 100 / (a[0, 3] >> 4), hw & 15, (hw >> 4) & 3, (hw >> 6) & 3, (hw >> 8) & 15, (hw >> 12) & 1, (hw >> 13) & 7, (hw >> 24) & 7))
    return out
for name, lanes in (("1 lane", [0]), ("32 lanes", list(range(32))), ("40 lanes", list(range(40))), ("48 lanes", list(range(48))), ("64 lanes", list(range(64)))):
    print(name)
    for p in pace(lanes):
        print("   %.3f us/step  wave %d simd %d pipe %d cu %d sh %d se %d queue %d" % p)
