#!/usr/bin/env python3
"""Does the longest wave finish sooner when its 8x8 tile is spread over several waves?  The tile that holds the frame's
longest ray is traced through lt_trace_batch_kerr (a) as one wavefront of 64 rays, (b) as k wavefronts each holding 64/k
of the tile's rays in its first lanes and short filler rays in the rest."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "light-path-tracer_amd"))
import ltrace
size = 4096
fov = np.radians(40.0)
cam = ltrace.Camera(size, size, fov, fov, 0.0, 0.0, 50.0, np.pi / 2)
alpha, theta, cols = ltrace.pixel_angles(cam)
def run(al, th, rf):
    n = al.size
    fa, w = np.empty(n), np.empty(n, dtype=np.int64); ev = np.empty(n, dtype=np.uint32)
    best = 1e9
    for _ in range(4):
        t0 = time.perf_counter()
        ltrace.trace_batch_kerr(1.0, 0.9, 50.0, al, th, np.pi / 2, 5000.0, rf, fa, w, integrator="rk4", precision=32, out_rhs_evals=ev)
        best = min(best, time.perf_counter() - t0)
    return best * 1e3, ev // 4
base, _ = run(np.full(64, 0.3), np.full(64, 1.0), np.zeros(64, np.uint8))
for (py, px) in ((402, 2046), (3292, 2055), (803, 2055)):
    y0, x0 = py // 8 * 8, px // 8 * 8
    al = alpha[y0:y0 + 8, x0:x0 + 8].astype(np.float64).ravel(); th = theta[y0:y0 + 8, x0:x0 + 8].ravel()
    rf = np.repeat(cols[x0:x0 + 8][None, :], 8, 0).astype(np.uint8).ravel()
    t1, st = run(al, th, rf)
    print(f"tile of pixel ({py},{px}): steps max {st.max()} p90 {np.percentile(st, 90):.0f} median {np.median(st):.0f}; one wave: {t1 - base:.3f} ms")
    for k in (2, 4, 8):
        per = 64 // k
        A = np.full(64 * k, 0.3); T = np.full(64 * k, 1.0); R = np.zeros(64 * k, np.uint8)
        for j in range(k):
            sl = slice(j * per, (j + 1) * per)
            A[64 * j:64 * j + per] = al[sl]; T[64 * j:64 * j + per] = th[sl]; R[64 * j:64 * j + per] = rf[sl]
        tk, _ = run(A, T, R)
        print(f"    spread over {k} waves of {per} tile rays: {tk - base:.3f} ms")
    j = int(np.argmax(st))
    ta, _ = run(np.full(64, al[j]), np.full(64, th[j]), np.full(64, rf[j], np.uint8))
    print(f"    its longest ray alone (replicated): {ta - base:.3f} ms")
