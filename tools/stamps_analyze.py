#!/usr/bin/env python3
"""Analyse LT_STAMPS_FILE output: one stamp per 8x8 tile (direct schedule: the wavefront that traced it, from taking the
tile to storing its last ray) or per wavefront (queue schedule): concurrency over time, duration vs steps."""
import sys
import numpy as np
a = np.fromfile(sys.argv[1], dtype=np.uint32).reshape(-1, 4)
t0 = a[:, 0].astype(np.int64); dur = a[:, 1].astype(np.int64)
t0 = (t0 - t0.min()) & 0xffffffff
t1 = t0 + dur
steps = a[:, 3] >> 4
xcc = a[:, 3] & 0xf
cyc = a[:, 2].astype(np.int64)   # shader-clock cycles of the wave's lifetime (s_memtime)
T = t1.max()
print(f"waves {len(a)}  kernel span {T/100:.1f} us (100 MHz ticks)  mean wave dur {dur.mean()/100:.1f} us  max {dur.max()/100:.1f} us")
print("ticks per step: median", np.median(dur / np.maximum(steps, 1)), " lone-ish (last 1% finishing)", np.median((dur / np.maximum(steps, 1))[np.argsort(t1)[-len(a)//100:]]))
# concurrency timeline in 20 bins
edges = np.linspace(0, T, 21)
for i in range(20):
    lo, hi = edges[i], edges[i + 1]
    ov = np.clip(np.minimum(t1, hi) - np.maximum(t0, lo), 0, None).sum() / (hi - lo)
    started = ((t0 >= lo) & (t0 < hi)).sum()
    print(f"  t={lo/100:8.1f}-{hi/100:8.1f} us  resident waves {ov:8.1f}  ({ov/1024:.2f}/SIMD)  started {started}")
order = np.argsort(-dur)[:8]
for i in order:
    print(f"  long wave {i}: start {t0[i]/100:.1f} us dur {dur[i]/100:.1f} us steps {steps[i]} us/step {dur[i]/100/max(steps[i],1):.4f} "
          f"cycles/step {cyc[i]/max(steps[i],1):.0f} clock {cyc[i]/max(dur[i],1)*100:.0f} MHz xcc {xcc[i]}")
print("waves per xcc:", np.bincount(xcc, minlength=8))
print(f"clock seen by waves: median {np.median(cyc / np.maximum(dur, 1)) * 100:.0f} MHz, lifetime-weighted {cyc.sum() / max(dur.sum(), 1) * 100:.0f} MHz")
