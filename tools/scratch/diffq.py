import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.getcwd(), "light-path-tracer_amd"))
import ltrace
size = 2048
fov = np.radians(40.0)
cam = ltrace.Camera(size, size, fov, fov, 0.0, 0.0, 50.0, np.pi / 2)
met = ltrace.Metric(1, 0, 1.0, 0.9)
a = ltrace.render(cam, met, ltrace.default_opts(precision=32, schedule="direct"), want=("fa", "status", "steps"))
b = ltrace.render(cam, met, ltrace.default_opts(precision=32, schedule="queue"), want=("fa", "status", "steps"))
d = ~((a["fa"] == b["fa"]) | (np.isnan(a["fa"]) & np.isnan(b["fa"])))
print("differing fa:", d.sum(), "steps differ:", (a["steps"] != b["steps"]).sum(), "status differ:", (a["status"] != b["status"]).sum())
ys, xs = np.nonzero(d)
for y, x in list(zip(ys, xs))[:12]:
    print(y, x, a["fa"][y, x], b["fa"][y, x], a["steps"][y, x], b["steps"][y, x], a["status"][y, x], b["status"][y, x])
print("x range", xs.min() if len(xs) else None, xs.max() if len(xs) else None, "steps of differing rays: min", a["steps"][d].min() if d.any() else None, "median", np.median(a["steps"][d]) if d.any() else None)
