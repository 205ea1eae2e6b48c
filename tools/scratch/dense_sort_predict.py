#!/usr/bin/env python3
"""How well does a loose-tolerance pass predict the step attempts of the tight one?  Lane-utilisation bound of the dense
kernel (sum of attempts / 64 x sum over waves of the longest) when tracks are grouped by the predictor."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "light-path-tracer_amd"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import ltrace
from dense_bench import states
n = 1 << 18
s0 = states(n, 0.9)
met = ltrace.Metric(1, 0, 1.0, 0.9)
def attempts(**kw):
    t, y, cnt, st, nfev = ltrace.integrate_dense(met, s0, ltrace.default_dense_opts(max_points=8, **kw))
    return ((nfev - 2) // 6).astype(np.int64), st
tight, st = attempts()
def bound(order):
    w = tight[order].reshape(-1, 64)
    return w.sum() / (64 * w.max(axis=1).sum())
print(f"tight pass: mean attempts {tight.mean():.1f}; bound as given {bound(np.arange(n)):.3f}; sorted by itself {bound(np.argsort(tight, kind='stable')):.3f}")
for rtol, atol in ((1e-3, 1e-5), (1e-4, 1e-6), (1e-5, 1e-7), (1e-6, 1e-8)):
    loose, _ = attempts(rtol=rtol, atol=atol)
    o = np.argsort(loose, kind="stable")
    print(f"predictor rtol {rtol:g}: mean attempts {loose.mean():.1f} ({loose.mean() / tight.mean():.2f} of the tight pass), corr {np.corrcoef(loose, tight)[0, 1]:.3f}, bound when sorted by it {bound(o):.3f}")
    # with status as secondary information
    o2 = np.lexsort((loose, st))
    print(f"     sorted by (ending, predictor): {bound(o2):.3f}")
print("predictor = affine parameter at the end of a pass with a free step size (max_step 1e9):")
for rtol, atol in ((1e-2, 1e-4), (1e-3, 1e-5), (1e-4, 1e-6)):
    t, y, cnt, st2, nfev = ltrace.integrate_dense(met, s0, ltrace.default_dense_opts(max_points=8, rtol=rtol, atol=atol, max_step=1e9))
    lam_end = t[np.arange(n), np.minimum(cnt, 8) - 1]
    a2 = ((nfev - 2) // 6).astype(np.int64)
    o = np.argsort(lam_end, kind="stable")
    o2 = np.lexsort((lam_end, st2))
    print(f"   rtol {rtol:g}: mean attempts of the pass {a2.mean():.1f} ({a2.mean() / tight.mean():.3f} of the tight pass), endings agree {np.mean(st2 == st):.4f}, "
          f"corr {np.corrcoef(lam_end, tight)[0, 1]:.3f}, bound sorted by lambda_end {bound(o):.3f}, by (ending, lambda_end) {bound(o2):.3f}")
print("predictor = least-squares fit of the tight attempts on (lambda_end, attempts of the free-step pass, ending):")
for rtol, atol in ((1e-3, 1e-5), (1e-4, 1e-6)):
    t, y, cnt, st2, nfev = ltrace.integrate_dense(met, s0, ltrace.default_dense_opts(max_points=8, rtol=rtol, atol=atol, max_step=1e9))
    lam_end = t[np.arange(n), np.minimum(cnt, 8) - 1]
    a2 = ((nfev - 2) // 6).astype(np.float64)
    X = np.stack([np.ones(n), lam_end, a2, (st2 == 1).astype(float), a2 * a2, lam_end * (st2 == 1)], axis=1)
    half = n // 2
    coef, *_ = np.linalg.lstsq(X[:half], tight[:half].astype(float), rcond=None)
    pred = X @ coef
    print(f"   rtol {rtol:g}: corr {np.corrcoef(pred[half:], tight[half:])[0, 1]:.3f}, bound sorted by the fit {bound(np.argsort(pred, kind='stable')):.3f}; coef {np.round(coef, 3)}")
