#!/usr/bin/env python3
"""Tiles whose time per step is far above the median (LT_STAMPS_FILE of a tile-hand-out launch): how many, where in the queue, when"""
import sys
import numpy as np
a = np.fromfile(sys.argv[1], dtype=np.uint32).reshape(-1, 4)
t0 = a[:, 0].astype(np.int64); dur = a[:, 1].astype(np.int64)
t0 = (t0 - t0.min()) & 0xffffffff
steps = np.maximum(a[:, 3] >> 4, 1)
cyc = a[:, 2].astype(np.int64)
ups = dur / 100 / steps
med = np.median(ups)
print(f"tiles {len(a)}; us per step of the longest ray: median {med:.3f}, p90 {np.percentile(ups, 90):.3f}, p99 {np.percentile(ups, 99):.3f}, p99.9 {np.percentile(ups, 99.9):.3f}, max {ups.max():.3f}")
for f in (1.5, 2.0, 2.5, 3.0):
    m = ups > f * med
    print(f"  > {f} x median: {m.sum()} tiles, {dur[m].sum() / dur.sum() * 100:.2f} % of all tile time; steps median {np.median(steps[m]) if m.any() else 0:.0f}; "
          f"queue position median {np.median(np.nonzero(m)[0]) if m.any() else 0:.0f}; start time median {np.median(t0[m]) / 100 if m.any() else 0:.0f} us")
m = ups > 2.0 * med
idx = np.nonzero(m)[0]
print("examples (tile, start us, dur us, steps, cycles/step, clock MHz):")
for i in idx[:: max(1, len(idx) // 15)][:15]:
    print(f"   {i:7d}  {t0[i] / 100:8.1f}  {dur[i] / 100:8.1f}  {steps[i]:5d}  {cyc[i] / steps[i]:7.0f}  {cyc[i] / max(dur[i], 1) * 100:5.0f}")
# time structure: slow tiles vs time
T = (t0 + dur).max()
for lo in np.linspace(0, T, 11)[:-1]:
    hi = lo + T / 10
    sel = (t0 >= lo) & (t0 < hi)
    print(f"  started in [{lo / 100:7.0f}, {hi / 100:7.0f}) us: {sel.sum():6d} tiles, median us/step {np.median(ups[sel]):.3f}, p99 {np.percentile(ups[sel], 99):.3f}, mean steps {steps[sel].mean():.0f}")
