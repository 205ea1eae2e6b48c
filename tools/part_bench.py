#!/usr/bin/env python3
"""What one rank of an N-GPU run does, measured on ONE GPU under benchmark conditions: for every partition p of n,
back-to-back device-resident frames of that partition (no host copies in between, so the clocks stay up -- unlike
tools/part_times.py, which goes through the host-pointer API and lets the GPU idle between launches).
The N-GPU frame time is the max over p, plus the gather.   usage: part_bench.py [size] [rk4|dp45] [frames] [spin a]"""
import os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "light-path-tracer_amd"))
import ltrace
size = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
integ = sys.argv[2] if len(sys.argv) > 2 else "rk4"
frames = int(sys.argv[3]) if len(sys.argv) > 3 else 10
spin = float(sys.argv[4]) if len(sys.argv) > 4 else 0.9
dev = torch.device("cuda:0")
fov = np.radians(40.0)
cam = ltrace.Camera(size, size, fov, fov, 0.0, 0.0, 50.0, np.pi / 2)
met = ltrace.Metric(1, 0, 1.0, spin)
stream = torch.cuda.current_stream()
# bring the chip to its loaded clocks first (a cold box runs its first second faster, then settles): ~1.5 s of full frames
_w = torch.empty((size, size, 4), dtype=torch.uint8, device=dev)
_o = ltrace.default_opts(integrator=integ, precision=32 if integ == "rk4" else 64)
_o.stream = stream.cuda_stream
_t = time.perf_counter()
while time.perf_counter() - _t < 1.5:
    ltrace.render_dev(cam, met, _o, d_rgba=_w.data_ptr())
    torch.cuda.synchronize()
del _w
for n in (1, 2, 4, 8):
    res = []
    for p in range(n):
        rows = ltrace.local_rows(size, 16, n, p)
        rgba = torch.empty((rows, size, 4), dtype=torch.uint8, device=dev)
        o = ltrace.default_opts(integrator=integ, precision=32 if integ == "rk4" else 64, n_parts=n, part=p, row_block=16, timing=1)
        o.stream = stream.cuda_stream
        step = lambda: ltrace.render_dev(cam, met, o, d_rgba=rgba.data_ptr())
        for _ in range(3):
            step()
        torch.cuda.synchronize(); ltrace.timing_collect()
        t0 = time.perf_counter()
        for _ in range(frames):
            step()
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) / frames * 1e3
        tm = ltrace.timing_collect()
        res.append((tm["integrate_ms"] / frames, wall))
    print(f"n_parts={n}: integrate ms per part = {[round(a, 2) for a, _ in res]}  frame ms per part = {[round(b, 2) for _, b in res]}"
          f"  -> slowest rank {max(b for _, b in res):.2f} ms per frame = {size * size / max(b for _, b in res) / 1e3:.0f} Mrays/s before the gather")
