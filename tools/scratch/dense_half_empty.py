#!/usr/bin/env python3
"""Do half-empty wavefronts cost more than full ones when the SIMDs are shared (dense-track kernel, 2 waves per SIMD)?
Batch A: every track the same long track.  Batch B: the same, but lanes 32..63 (or all but one lane) of every wave hold a
track that ends at once.  Same number of waves and of wave-instructions; LT_DENSE_NOSTORE=1 keeps the record stores out."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tools"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "light-path-tracer_amd"))
import ltrace
from dense_bench import states
n, mp, a = 1 << 18, 64, 0.9
dev = torch.device("cuda:0")
base = states(4, a)[1]
long_ = np.tile(base, (n, 1))
short = base.copy(); short[1] = 1.46; short[5] = -1.0
def batch(keep, keep_odd=None):
    s = long_.copy()
    lane = np.arange(n) % 64
    wave = np.arange(n) // 64
    drop = ~np.isin(lane, keep)
    if keep_odd is not None:
        drop = np.where(wave % 2 == 1, ~np.isin(lane, keep_odd), drop)
    s[drop] = short
    return s
t = torch.empty((n, mp), dtype=torch.float64, device=dev); y = torch.empty((n, mp, 8), dtype=torch.float64, device=dev)
cnt = torch.zeros(n, dtype=torch.int32, device=dev); st = torch.zeros(n, dtype=torch.int8, device=dev); nf = torch.zeros(n, dtype=torch.int32, device=dev)
stream = torch.cuda.current_stream()
o = ltrace.default_dense_opts(max_points=mp, stream=stream.cuda_stream)
met = ltrace.Metric(ltrace.METRIC_KERR, 0, 1.0, a)
F, S16, S1 = list(range(64)), list(range(16)), [0]
cases = [("all waves 64", F, None), ("all waves 16", S16, None), ("all waves 1", S1, None), ("even waves 64, odd waves 1", F, S1),
         ("even waves 64, odd waves 16", F, S16), ("even waves 16, odd waves 1", S16, S1), ("all waves 64", F, None)]
for name, keep, keep_odd in cases:
    s0 = torch.from_numpy(batch(keep, keep_odd)).to(dev)
    run = lambda: ltrace.integrate_dense_dev(met, o, s0.data_ptr(), n, t.data_ptr(), y.data_ptr(), cnt.data_ptr(), st.data_ptr(), nf.data_ptr())
    run(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    for _ in range(5): run()
    e1.record(stream); torch.cuda.synchronize()
    print(f"{name:24s} {e0.elapsed_time(e1) / 5:8.3f} ms   evals: long lane {int(nf[0])}, last lane {int(nf[63])}")
