#!/usr/bin/env python3
"""Host-pointer lt_render at 4096^2, RGBA8 destination only: pinned (lt_host_alloc) vs pageable destination, the
same buffer passed again every frame.  What `python image_lens.py` pays per frame."""
import ctypes as C, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "light-path-tracer_amd"))
import ltrace
size = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
fov = np.radians(40.0)
cam = ltrace.Camera(size, size, fov, fov, 0.0, 0.0, 50.0, np.pi / 2)
met = ltrace.Metric(1, 0, 1.0, 0.9)
o = ltrace.default_opts(precision=32)
for name in ("pinned", "pageable"):
    rgba = ltrace.pinned_empty((size, size, 4), np.uint8, strict=True) if name == "pinned" else np.zeros((size, size, 4), np.uint8)
    st = ltrace.Stats()
    call = lambda: ltrace._check(ltrace.load().lt_render(C.byref(cam), C.byref(met), C.byref(o), None, 3, None, None, None, None, None,
                                                         C.c_void_p(rgba.ctypes.data), C.byref(st)))
    call(); call()
    ts = []
    for _ in range(8):
        t0 = time.perf_counter(); call(); ts.append((time.perf_counter() - t0) * 1e3)
    t0 = time.perf_counter(); s = int(rgba[::7].sum()); tr = (time.perf_counter() - t0) * 1e3
    print(f"{name:8s} dst: median {np.median(ts):7.2f} ms  min {min(ts):7.2f}  (kernels {st.prologue_ms + st.integrate_ms + st.epilogue_ms:.2f} ms); "
          f"host read of 1/7 of the frame {tr:.1f} ms (checksum {s})")
