#!/usr/bin/env python3
"""Wave stamps of one partition's integrate kernel: LT_STAMPS_FILE=f python tools/scratch/stamp_part.py size n_parts part"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "light-path-tracer_amd"))
import ltrace
size, n, p = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
fov = np.radians(40.0)
cam = ltrace.Camera(size, size, fov, fov, 0.0, 0.0, 50.0, np.pi / 2)
met = ltrace.Metric(1, 0, 1.0, 0.9)
for _ in range(3):
    out = ltrace.render(cam, met, ltrace.default_opts(precision=32, n_parts=n, part=p), want=("steps",))
print("integrate ms", out["stats"]["integrate_ms"], "max steps", out["steps"].max())
