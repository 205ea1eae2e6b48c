#!/usr/bin/env python3
"""dump (mode 'dump OUT.npz') the outputs of a small Kerr frame, or compare two dumps (mode 'cmp A B')"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "light-path-tracer_amd"))
if sys.argv[1] == "dump":
    import ltrace
    W = H = 256
    fov = np.radians(40.0)
    cam = ltrace.Camera(W, H, fov, fov, 0.0, 0.0, 50.0, np.pi / 2)
    o = ltrace.default_opts(precision=32, schedule="direct")
    out = ltrace.render(cam, ltrace.Metric(1, 0, 1.0, 0.9), o, want=("fa", "winding", "status", "steps"))
    np.savez(sys.argv[2], **{k: np.asarray(v) for k, v in out.items() if k in ("fa", "winding", "status", "steps")})
else:
    a, b = np.load(sys.argv[2]), np.load(sys.argv[3])
    for k in a.files:
        x, y = a[k], b[k]
        neq = ~((x == y) | (np.isnan(x.astype(float)) & np.isnan(y.astype(float))))
        print(k, "differs in", int(neq.sum()), "of", x.size)
        if neq.any() and k == "fa":
            idx = np.argwhere(neq)[:8]
            for i in idx: print("   ", tuple(i), x[tuple(i)], y[tuple(i)], "steps", a["steps"][tuple(i)], b["steps"][tuple(i)])
