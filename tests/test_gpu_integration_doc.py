"""The binding stubs printed in INTEGRATION.md (what a maintainer of the reference would paste into its metrics.py /
geodesic_tracer.py) are executed here against the built library, so that the document cannot drift from the C-ABI."""
import os
import re

import numpy as np
import pytest

import geodesic_tracer as gt
import ltrace
import metrics

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _stub(section_heading):
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    block = re.search(r"```python\n(.*?)```", text[text.index(section_heading):], re.S).group(1)
    block = block.replace('C.CDLL("/path/to/libltrace_hip.so")', f'C.CDLL({ltrace.LIB_PATH!r})')
    ns = {"np": np}
    exec(compile(block, "INTEGRATION.md", "exec"), ns)
    return ns


def test_batch_binding_stub_of_section_b():
    ns = _stub("## B. Keep the reference's scripts")
    rng = np.random.default_rng(1)
    n = 3000
    al, th, rf = rng.uniform(0.0, 0.4, n), rng.uniform(-np.pi, np.pi, n), rng.random(n) < 0.2
    fa1, w1 = np.full(n, np.nan), np.zeros(n, dtype=np.int64)
    ns["_trace_rays_batch_schwarzschild"](1.0, 2.0, 50.0, al, 50.0, 0.05, fa1, w1)
    fa2, w2 = np.full(n, np.nan), np.zeros(n, dtype=np.int64)
    metrics.Schwarzschild(1.0).trace_rays_batch(50.0, al, fa2, w2)
    assert np.array_equal(w1, w2) and np.array_equal(fa1, fa2, equal_nan=True)
    # Kerr: in place into slices of larger buffers, as image_lens.py:255-258 calls it
    big_fa, big_w = np.full(2 * n, np.nan), np.zeros(2 * n, dtype=np.int64)
    ns["_trace_rays_batch_kerr"](1.0, 0.9, 1.4358898943540672, 50.0, al, th, np.pi / 2, 5000.0, rf, big_fa[n:], big_w[n:])
    fa3, w3 = np.full(n, np.nan), np.zeros(n, dtype=np.int64)
    metrics.Kerr(1.0, 0.9).trace_rays_batch(50.0, al, th, np.pi / 2, rf, fa3, w3)   # the plugin default == the stub's `2, 64`
    assert np.all(np.isnan(big_fa[:n])) and np.all(big_w[:n] == 0)
    assert np.array_equal(big_w[n:], w3) and np.array_equal(big_fa[n:], fa3, equal_nan=True)
    assert 0.5 < np.isfinite(fa3).mean() < 0.95


def test_dense_binding_stub_of_section_c():
    ns = _stub("## C. Dense trajectories")
    for met in (metrics.Schwarzschild(1.0), metrics.Kerr(1.0, 0.9)):
        degs = [2, 5.97, 8, 15]
        res = ns["integrate_geodesics"](met, [met.initial_conditions(50.0, np.radians(d)) for d in degs])
        for d, (t, y, outcome) in zip(degs, res):
            sol, oc = gt.trace_ray(met, 50.0, np.radians(d))
            assert outcome == oc and t.shape == sol.t.shape
            assert np.max(np.abs(y - sol.y) / (1 + np.abs(sol.y))) < 1e-8
