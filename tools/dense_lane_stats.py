#!/usr/bin/env python3
"""Where the dense-trajectory kernel's idle lanes come from: per-track step attempts of the dense_bench workload,
grouped 64 consecutive tracks to a wavefront -> sum(attempts) / (64 x max(attempts)) = the lane utilisation a
one-track-per-lane launch can reach at best."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "light-path-tracer_amd"))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import ltrace
from dense_bench import states
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 18
s0 = states(n, 0.9)
t, y, cnt, st, nfev = ltrace.integrate_dense(ltrace.Metric(1, 0, 1.0, 0.9), s0, ltrace.default_dense_opts(max_points=8))
att = ((nfev - 2) // 6).astype(np.int64)
w = att.reshape(-1, 64)
print(f"attempts per track: mean {att.mean():.1f}  p5 {np.percentile(att, 5):.0f}  median {np.median(att):.0f}  p95 {np.percentile(att, 95):.0f}  max {att.max()}")
print(f"captured {np.mean(st == 1):.3f} of tracks, mean attempts {att[st == 1].mean():.1f}; escaped mean attempts {att[st == 2].mean():.1f}")
print(f"one track per lane, 64 consecutive tracks per wave: utilisation bound sum / (64 max) = {w.sum() / (64 * w.max(axis=1).sum()):.3f}")
o = np.argsort(att)
ws = att[o].reshape(-1, 64)
print(f"same, tracks sorted by length first (what perfect grouping would give): {ws.sum() / (64 * ws.max(axis=1).sum()):.3f}")
srt = np.sort(w, axis=1)[:, ::-1]          # per wave, longest first
tot = srt[:, 0].sum()
for live in (1, 2, 4, 8, 12, 16, 32):
    # iterations of a wave during which at most `live` lanes are still running: from the (live+1)-th longest track's end to the longest's
    print(f"share of wave iterations with <= {live:2d} live lanes: {(srt[:, 0] - srt[:, live]).sum() / tot:.3f}")
