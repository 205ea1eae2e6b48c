"""GPU parity of the batched dense-trajectory path (lt_integrate_dense), through the C-ABI.

Checked against (i) the reference's own solve_ivp tracks (tests/golden/dense_tracks.npz, F10), (ii) the CPU
oracle (oracle/lt_oracle_dense.c, itself pinned by F10) on seeded batches, (iii) properties at scale:
every track ends on its event radius or at lambda_max, the Hamiltonian and the two cyclic momenta are
conserved, the batch does not depend on how tracks are grouped.

Tolerance (float64): points within 1e-8 of the reference's relative to 1 + |value| (the GPU evaluates an
algebraically equivalent right-hand side with fused multiply-adds, which moves adaptive step sizes by
~1e-11); the NUMBER of points and of right-hand-side evaluations must be equal.
"""
import os
import time

import numpy as np
import pytest

import geodesic_tracer as gt
import ltrace
import metrics
from oracle import oracle

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _gold():
    return np.load(os.path.join(GOLD, "dense_tracks.npz"), allow_pickle=False)


def _metric(M, a, kerr):
    return ltrace.Metric(ltrace.METRIC_KERR if kerr else ltrace.METRIC_SCHWARZSCHILD, 0, float(M), float(a))


def test_rhs8_probe_matches_reference():
    g = _gold()
    for M, a in ((1.0, 0.0), (1.0, 0.9), (1.0, 0.99)):
        sel = (g["rhs_M_a"][:, 1] == a)
        out = ltrace.rhs8_probe(_metric(M, a, a != 0), g["rhs_state"][sel])
        exp = g["rhs_out"][sel]
        scale = np.max(np.abs(exp), axis=1, keepdims=True)
        assert np.max(np.abs(out - exp) / scale) < 1e-12


def test_tracks_match_reference_solve_ivp():
    g = _gold()
    off = g["offsets"]
    worst = 0.0
    for i in range(len(off) - 1):
        M, a = g["M_a"][i]
        lam, r_in, r_out = g["stops"][i]
        o = ltrace.default_dense_opts(lambda_max=lam, r_stop_inner=r_in, r_stop_outer=r_out, max_points=1024)
        t, y, count, status, nfev = ltrace.integrate_dense(_metric(M, a, g["metric_id"][i] > 0), g["state0"][i][None], o)
        gt_, gy = g["t"][off[i]:off[i + 1]], g["y"][:, off[i]:off[i + 1]]
        assert count[0] == len(gt_) and nfev[0] == g["nfev"][i], f"track {i}: {count[0]} vs {len(gt_)} points"
        assert (status[0] in (1, 2)) == (g["ivp_status"][i] == 1)
        m = count[0]
        np.testing.assert_allclose(t[0, :m], gt_, rtol=0, atol=1e-7)
        err = np.max(np.abs(y[0, :m].T - gy) / (1 + np.abs(gy)))
        worst = max(worst, err)
        assert err < 1e-8, f"track {i}"
    print(f"worst relative point difference vs solve_ivp: {worst:.2e}")


@pytest.mark.parametrize("kerr,a", [(False, 0.0), (True, 0.9), (True, 0.99)])
def test_batch_matches_oracle(kerr, a):
    """A seeded batch of 300 tracks in ONE launch against the oracle track by track."""
    rng = np.random.default_rng(5)
    met = metrics.Kerr(1.0, a) if kerr else metrics.Schwarzschild(1.0)
    n = 300
    alphas = rng.uniform(0.0, 0.35, n)
    thetas = rng.uniform(0, 2 * np.pi, n) if kerr else np.zeros(n)
    s0 = np.array([met.initial_conditions(50.0, al, th) for al, th in zip(alphas, thetas)])
    o = ltrace.default_dense_opts(max_points=700)
    t, y, count, status, nfev = ltrace.integrate_dense(_metric(1.0, a, kerr), s0, o)
    r_plus = 1.0 + np.sqrt(1.0 - a * a)
    same_steps, errs, cpu_s = 0, [], 0.0
    for i in range(n):
        t0 = time.perf_counter()
        ot, oy, ost, onfev = oracle.integrate_dense(int(kerr), 1.0, a, s0[i], 1000.0, 1.01 * r_plus, 2 * s0[i][1])
        cpu_s += time.perf_counter() - t0
        assert ost == status[i]
        if len(ot) == count[i] and onfev == nfev[i]:
            same_steps += 1
            m = count[i]
            errs.append(np.max(np.abs(y[i, :m].T - oy) / (1 + np.abs(oy))))
        else:  # an accept / reject decision within rounding of err = 1: the end point still agrees
            assert abs(int(count[i]) - len(ot)) <= 2
            np.testing.assert_allclose(y[i, min(count[i], 700) - 1], oy[:, -1], rtol=1e-6, atol=1e-6)
    assert same_steps >= n - 3, f"{n - same_steps} tracks took a different step sequence"
    errs = np.array(errs)
    print(f"a={a}: median {np.median(errs):.1e}  p99 {np.quantile(errs, 0.99):.1e}  max {errs.max():.1e};"
          f"  oracle (C, one host core): {n / cpu_s:.0f} tracks/s")
    # tracks that graze the photon orbit amplify last-bit differences exponentially: budget the tail separately
    assert np.median(errs) < 1e-9 and np.quantile(errs, 0.98) < 1e-8 and errs.max() < 1e-5


def test_large_batch_properties():
    """65 536 Kerr tracks: endings, conservation laws, independence of the grouping."""
    a, n = 0.9, 65536
    rng = np.random.default_rng(11)
    met = metrics.Kerr(1.0, a)
    alphas = rng.uniform(0.01, 0.4, n)
    thetas = rng.uniform(0, 2 * np.pi, n)
    s0 = np.array([met.initial_conditions(50.0, al, th) for al, th in zip(alphas, thetas)])
    MP = 384
    o = ltrace.default_dense_opts(max_points=MP)
    lm = _metric(1.0, a, True)
    t, y, count_full, status, nfev = ltrace.integrate_dense(lm, s0, o)
    assert np.all(status >= 1), "at r_obs = 50 every track ends on an event well inside lambda_max = 1000"
    assert np.all(count_full >= 30) and np.mean(count_full > MP) < 1e-3  # a few near-critical tracks orbit for long
    count = np.minimum(count_full, MP)
    idx = np.arange(n)
    last = y[idx, count - 1]                            # (n, 8) final points (kept even when truncated)
    r_in = 1.01 * (1 + np.sqrt(1 - a * a))
    target = np.where(status == 1, r_in, 100.0)
    assert np.max(np.abs(last[:, 1] - target)) < 1e-9   # the event radius, located to 4 eps by Brent's method
    assert np.all(last[:, 4] == -1.0) and np.all(last[:, 7] == s0[:, 7])  # cyclic momenta: exactly constant
    # null condition along the track: H = g^{mu nu} p_mu p_nu / 2 stays at its start value (~0) to the tolerance
    def hamiltonian(p):
        r, th, pt, pr, pth, pph = p[:, 1], p[:, 2], p[:, 4], p[:, 5], p[:, 6], p[:, 7]
        S = r * r + a * a * np.cos(th) ** 2; D = r * r - 2 * r + a * a; s2 = np.sin(th) ** 2
        A = (r * r + a * a) ** 2 - a * a * D * s2
        terms = np.stack([-A / (S * D) * pt * pt, -4 * a * r / (S * D) * pt * pph, D / S * pr * pr, pth * pth / S,
                          (D - a * a * s2) / (S * D * s2) * pph * pph])
        return np.abs(terms.sum(0)) / np.abs(terms).sum(0)   # |2H| relative to the size of its terms
    mid = y[idx, count // 2]
    h_mid, h_last = hamiltonian(mid), hamiltonian(last)
    print(f"relative null-condition residual: mid-track max {h_mid.max():.1e}, end max {h_last.max():.1e}")
    assert h_mid.max() < 1e-6 and h_last.max() < 1e-6
    # times strictly increase along every track
    assert np.all(np.diff(t, axis=1)[np.arange(MP - 1)[None, :] < (count - 1)[:, None]] > 0)
    # grouping: a permuted batch gives bit-identical tracks
    perm = rng.permutation(n)[:4096]
    t2, y2, c2, st2, nf2 = ltrace.integrate_dense(lm, s0[perm], o)
    assert np.array_equal(c2, count_full[perm]) and np.array_equal(st2, status[perm]) and np.array_equal(nf2, nfev[perm])
    for j in (0, 17, 4095):
        m = min(c2[j], MP)
        assert np.array_equal(y2[j, :m], y[perm[j], :m]) and np.array_equal(t2[j, :m], t[perm[j], :m])


def _dense_states(n, a, seed):
    rng = np.random.default_rng(seed)
    met = metrics.Kerr(1.0, a) if a else metrics.Schwarzschild(1.0)
    return np.array([met.initial_conditions(50.0, al, th) for al, th in zip(rng.uniform(0.01, 0.4, n), rng.uniform(0, 2 * np.pi, n))])


@pytest.mark.parametrize("a,n", [(0.9, 70000), (0.0, 5001)])
def test_length_binned_launch_is_byte_identical(a, n):
    """lt_dense_opts.length_binning orders the LAUNCH (tracks longest-first by a predicted length), never the output:
    every record, count, ending and nfev equals the launch in caller order bit for bit.  n is ragged on purpose (not a
    multiple of 64 nor of the 1024 tracks a sorting workgroup places)."""
    s0 = _dense_states(n, a, 21)
    lm = _metric(1.0, a, a != 0)
    MP = 320
    plain = ltrace.integrate_dense(lm, s0, ltrace.default_dense_opts(max_points=MP, length_binning=-1))
    binned = ltrace.integrate_dense(lm, s0, ltrace.default_dense_opts(max_points=MP, length_binning=1))
    for name, x, z in zip(("count", "status", "nfev"), plain[2:], binned[2:]):
        assert np.array_equal(x, z), name
    m = np.minimum(plain[2], MP)
    live = np.arange(MP)[None, :] < m[:, None]                    # slots past a track's last point are never written
    assert np.array_equal(plain[0][live], binned[0][live]) and np.array_equal(plain[1][live], binned[1][live])
    # twice in a row on the same stream: the workspace (histogram, cursors, queue heads) is re-armed per call
    again = ltrace.integrate_dense(lm, s0, ltrace.default_dense_opts(max_points=MP, length_binning=1))
    assert np.array_equal(again[3], plain[3]) and np.array_equal(again[1][live], plain[1][live])


def test_length_predictor_matches_its_cpu_twin_and_orders_the_launch():
    """The keys the binned launch sorts by (float32 loose-tolerance pass on the GPU) against the predictor's CPU twin
    (oracle.dense_predict_length, float64), and what the order they give is worth: the lane-utilisation bound
    sum(attempts) / (64 x sum over wavefronts of the longest) of one track per lane rises from ~0.62 in caller order
    to >= 0.85 (1.0 = sorted by the true length)."""
    a, n = 0.9, 16384
    s0 = _dense_states(n, a, 33)
    lm = _metric(1.0, a, True)
    key = ltrace.dense_predict_lengths(lm, s0).astype(np.int64)
    t, y, cnt, st, nfev = ltrace.integrate_dense(lm, s0, ltrace.default_dense_opts(max_points=8, length_binning=-1))
    attempts = (nfev.astype(np.int64) - 2) // 6
    r_in = 1.01 * (1 + np.sqrt(1 - a * a))
    twin = np.array([oracle.dense_predict_length(1, 1.0, a, s0[i], 1000.0, r_in, 100.0)[0] for i in range(2048)])
    d = np.abs(key[:2048] - twin)
    assert np.quantile(d, 0.95) <= 0.05 * twin.mean() and np.median(d) <= 0.02 * twin.mean(), (np.median(d), np.quantile(d, 0.95))
    assert np.corrcoef(key, attempts)[0, 1] > 0.95

    def bound(order):
        w = attempts[order].reshape(-1, 64)
        return w.sum() / (64 * w.max(axis=1).sum())
    as_given, by_key = bound(np.arange(n)), bound(np.concatenate([w0 + np.argsort(-key[w0:w0 + 2048], kind="stable") for w0 in range(0, n, 2048)]))
    print(f"lane-utilisation bound: caller order {as_given:.3f}, ordered by the predicted length inside windows of 2048 {by_key:.3f}, "
          f"by the true length {bound(np.argsort(-attempts)):.3f}; predicted / true attempts {key.mean() / attempts.mean():.3f}")
    assert as_given < 0.70 and by_key >= 0.85


def test_length_predictor_keys_do_not_depend_on_the_launch():
    """The predictor's lanes are refilled from a queue once a batch exceeds what the chip holds at once (k_dense_predict:
    a lane whose track has ended takes the next one): every track's key must be what it is when the track runs in a
    wavefront of its own block -- 600 000 tracks through the queue against the same tracks in batches that fit the chip,
    key for key, and with other refill settings (one lane at a time; chunks that do not divide the batch)."""
    import os
    import subprocess
    import sys
    a, n = 0.9, 600_000 + 37
    base = _dense_states(4096, a, 7)
    rng = np.random.default_rng(11)
    s0 = base[rng.integers(0, 4096, n)]                      # tracks of every length, in no order
    s0[:, 5] *= 1.0 + 1e-6 * rng.standard_normal(n)          # (not 4096 distinct keys only)
    lm = _metric(1.0, a, True)
    queued = ltrace.dense_predict_lengths(lm, s0)
    small = np.concatenate([ltrace.dense_predict_lengths(lm, s0[i:i + 65536]) for i in range(0, n, 65536)])
    assert np.array_equal(queued, small)
    assert queued.min() > 0 and len(np.unique(queued)) > 50
    # other hand-out parameters: the environment is read once per process, so in a child
    code = ("import sys, numpy as np; sys.path.insert(0, %r); import ltrace; s0 = np.load(sys.argv[1]); "
            "np.save(sys.argv[2], ltrace.dense_predict_lengths(ltrace.Metric(ltrace.METRIC_KERR, 0, 1.0, %r), s0))"
            % (os.path.dirname(ltrace.__file__), a))
    import tempfile
    with tempfile.TemporaryDirectory() as tmp:
        np.save(os.path.join(tmp, "s0.npy"), s0)
        for refill, chunk in ((1, 64), (64, 48), (8, 1000)):
            env = dict(os.environ, LT_DENSE_PRED_REFILL=str(refill), LT_DENSE_PRED_CHUNK=str(chunk))
            out = os.path.join(tmp, f"k_{refill}_{chunk}.npy")
            subprocess.run([sys.executable, "-c", code, os.path.join(tmp, "s0.npy"), out], check=True, env=env, timeout=300)
            assert np.array_equal(np.load(out), queued), (refill, chunk)


def test_truncation_and_range_end():
    met = metrics.Schwarzschild(1.0)
    s0 = np.array([met.initial_conditions(50.0, np.radians(d)) for d in (8.0, 4.0)])
    lm = _metric(1.0, 0.0, False)
    full = ltrace.integrate_dense(lm, s0, ltrace.default_dense_opts(max_points=512))
    cut = ltrace.integrate_dense(lm, s0, ltrace.default_dense_opts(max_points=16))
    assert np.array_equal(full[2], cut[2]) and np.all(cut[2] > 16)       # counts report the complete record
    for i in range(2):
        assert np.array_equal(cut[1][i, :15], full[1][i, :15])             # first max_points - 1 points kept
        assert np.array_equal(cut[1][i, 15], full[1][i, full[2][i] - 1])   # last slot = final point
    t, y, count, status, nfev = ltrace.integrate_dense(lm, s0, ltrace.default_dense_opts(lambda_max=30.0))
    assert np.all(status == ltrace.TRACK_RANGE_END) and np.all(t[np.arange(2), count - 1] == 30.0)


def test_host_mirror_trace_rays_matches_scipy_path():
    """geodesic_tracer.trace_rays (GPU) against geodesic_tracer.trace_ray (scipy on the package's own 8-D equations)."""
    for met in (metrics.Schwarzschild(1.0), metrics.Kerr(1.0, 0.9)):
        degs = [0, 2, 5.97, 8, 15, 95]
        fan = gt.trace_rays(met, 50.0, np.radians(degs))
        for deg, (trk, outcome) in zip(degs, fan):
            sol, oc = gt.trace_ray(met, 50.0, np.radians(deg))
            if sol is None:
                assert trk is None and outcome == "invalid"
                continue
            assert outcome == oc and trk.nfev == sol.nfev and trk.t.shape == sol.t.shape and trk.status == sol.status
            assert np.max(np.abs(trk.y - sol.y) / (1 + np.abs(sol.y))) < 1e-8


def test_track_sol_is_solve_ivps_dense_output():
    """Track.sol(t) against `solution.sol(t)` of the scipy path (the reference asks solve_ivp for dense_output=True,
    geodesic_tracer.py:66): interior steps and the last stretch before the terminal event, which solve_ivp interpolates
    with the dense output of the whole step that contains the event."""
    rng = np.random.default_rng(4)
    for met in (metrics.Schwarzschild(1.0), metrics.Kerr(1.0, 0.9)):
        degs = [2, 5.97, 8, 15]
        fan = gt.trace_rays(met, 50.0, np.radians(degs))
        for deg, (trk, outcome) in zip(degs, fan):
            sol, oc = gt.trace_ray(met, 50.0, np.radians(deg))
            ts = np.concatenate([rng.uniform(sol.t[0], sol.t[-1], 40), sol.t[-1] - np.array([0.0, 1e-3, 0.3]) * (sol.t[-1] - sol.t[-2]),
                                 sol.t[[0, 1, 5]]])
            got, exp = trk.sol(ts), sol.sol(ts)
            assert got.shape == exp.shape == (8, ts.size)
            assert np.max(np.abs(got - exp) / (1 + np.abs(exp))) < 1e-8, deg
            assert np.allclose(trk.sol(float(sol.t[3])), sol.y[:, 3], rtol=0, atol=1e-8)
    cut = gt.integrate_geodesics(met, [met.initial_conditions(50.0, 0.2)], max_points=16)[0][0]
    assert cut.truncated
    with pytest.raises(RuntimeError):
        cut.sol(1.0)


def test_dense_error_codes():
    lm = _metric(1.0, 0.0, False)
    s0 = np.zeros((1, 8))
    with pytest.raises(ltrace.LtraceError) as e:
        ltrace.integrate_dense(_metric(1.0, 1.5, True), s0)
    assert e.value.code == ltrace.ERR_INVALID_ARG
    with pytest.raises(ltrace.LtraceError):
        ltrace.integrate_dense(lm, s0, ltrace.default_dense_opts(rtol=-1.0))
    out = ltrace.integrate_dense(lm, np.zeros((0, 8)))
    assert out[2].size == 0
