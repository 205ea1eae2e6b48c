#!/usr/bin/env python3
"""Can lt_render_dev be captured into a HIP graph by the caller (torch.cuda.CUDAGraph) and replayed?  Schwarzschild 1024^2
(launch-bound: three kernels of 20 / 67 / 64 us) and Kerr 2048^2; output compared with the plain call."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "light-path-tracer_amd"))
import ltrace
dev = torch.device("cuda:0")
fov = np.radians(40.0)
for name, size, met in (("schwarzschild 1024", 1024, ltrace.Metric(ltrace.METRIC_SCHWARZSCHILD, 0, 1.0, 0.0)), ("kerr 2048", 2048, ltrace.Metric(ltrace.METRIC_KERR, 0, 1.0, 0.9))):
    cam = ltrace.Camera(size, size, fov, fov, 0.0, 0.0, 50.0, np.pi / 2)
    rgba = torch.zeros((size, size, 4), dtype=torch.uint8, device=dev)
    ref = torch.zeros_like(rgba)
    side = torch.cuda.Stream(dev)
    o = ltrace.default_opts(precision=32)
    o.stream = side.cuda_stream
    with torch.cuda.stream(side):
        for _ in range(3):
            ltrace.render_dev(cam, met, o, d_rgba=ref.data_ptr())       # warm-up: workspaces exist before the capture
    side.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        o.stream = torch.cuda.current_stream(dev).cuda_stream
        ltrace.render_dev(cam, met, o, d_rgba=rgba.data_ptr())
    torch.cuda.synchronize()
    n = 200 if size == 1024 else 30
    for _ in range(3): g.replay()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): g.replay()
    torch.cuda.synchronize(); tg = (time.perf_counter() - t0) / n
    o.stream = side.cuda_stream
    with torch.cuda.stream(side):
        side.synchronize(); t0 = time.perf_counter()
        for _ in range(n): ltrace.render_dev(cam, met, o, d_rgba=ref.data_ptr())
        side.synchronize(); tp = (time.perf_counter() - t0) / n
    print(f"{name}: graph replay {tg * 1e3:.4f} ms per frame ({size * size / tg / 1e6:.0f} Mrays/s), plain calls {tp * 1e3:.4f} ms ({size * size / tp / 1e6:.0f} Mrays/s), identical output: {bool(torch.equal(rgba, ref))}")
