#!/usr/bin/env python3
"""Step-count population of a frame's longest rays (the serial chains): top-5 step counts of the 4096^2, 2048^2 and 8192^2 (a = 0.99)
frames and the sum of steps.  usage: longest_rays.py  (LTRACE_LIB selects a build)"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "light-path-tracer_amd"))
import ltrace
fov = np.radians(40.0)
for size, a in ((4096, 0.9), (2048, 0.9), (8192, 0.99)):
    cam = ltrace.Camera(size, size, fov, fov, 0.0, 0.0, 50.0, np.pi / 2)
    out = ltrace.render(cam, ltrace.Metric(1, 0, 1.0, a), ltrace.default_opts(precision=32), want=("steps",))
    st = out["steps"].ravel()
    top = np.sort(np.partition(st, -5)[-5:])[::-1]
    print(f"{size}^2 a={a}: top steps {top.tolist()}  rays > 2000 steps: {(st > 2000).sum()}  mean {st.mean():.2f}  integrate {out['stats']['integrate_ms']:.3f} ms", flush=True)
