#!/usr/bin/env python3
"""The end of the integrate kernel from LT_STAMPS_FILE: resident waves per 50 us over the last millisecond, and what the
last waves to finish were (queue position = wave index, start, duration, steps of their longest ray)."""
import sys
import numpy as np
a = np.fromfile(sys.argv[1], dtype=np.uint32).reshape(-1, 4)
t0 = a[:, 0].astype(np.int64); dur = a[:, 1].astype(np.int64)
t0 = (t0 - t0.min()) & 0xffffffff
t1 = t0 + dur
steps = a[:, 3] >> 4
T = t1.max()
print(f"waves {len(a)}, span {T / 100:.1f} us; waves are dispatched in queue order: wave index = queue position")
for lo in range(int(T) - 100000, int(T), 5000):
    hi = lo + 5000
    ov = np.clip(np.minimum(t1, hi) - np.maximum(t0, lo), 0, None).sum() / 5000
    st = ((t0 >= lo) & (t0 < hi)).sum()
    print(f"  t = T - {(T - lo) / 100:6.0f} us: resident {ov:7.1f} ({ov / 1024:.2f}/SIMD), started {st}")
last_start = t0.max()
print(f"last wave dispatched at T - {(T - last_start) / 100:.0f} us")
late = np.argsort(-t1)[:12]
for i in late:
    print(f"  wave {i:7d} (of {len(a)}): started T - {(T - t0[i]) / 100:7.1f} us, ran {dur[i] / 100:7.1f} us, {steps[i]} steps")
d_last = dur[t0 > last_start - 30000]
print(f"waves dispatched in the last 300 us of dispatching: n {len(d_last)}, duration mean {d_last.mean() / 100:.0f} us, p90 {np.percentile(d_last, 90) / 100:.0f}, max {d_last.max() / 100:.0f} us; steps mean {steps[t0 > last_start - 30000].mean():.0f}")
# idle slot-time after dispatch ended
cap = 1024 * 5
idle = sum(max(0.0, cap - np.clip(np.minimum(t1, lo + 1000) - np.maximum(t0, lo), 0, None).sum() / 1000) * 1000 for lo in range(int(last_start), int(T), 1000))
print(f"wave-slot time unused after the last dispatch: {idle / cap / 100:.0f} us of full-chip time")
xcc = a[:, 3] & 0xf
cyc = a[:, 2].astype(np.int64)
print("per XCC: waves, last dispatch, last end (us before T), sum of wave durations (ms), steps total (M), mean clock seen (MHz)")
for x in range(8):
    m = xcc == x
    print(f"  xcc {x}: {m.sum():6d}  last dispatch T - {(T - t0[m].max()) / 100:6.1f}  last end T - {(T - t1[m].max()) / 100:6.1f}  "
          f"wave time {dur[m].sum() / 1e5:8.1f} ms  max-steps sum {steps[m].sum() / 1e6:6.2f} M  clock {cyc[m].sum() / max(dur[m].sum(), 1) * 100:.0f} MHz")
