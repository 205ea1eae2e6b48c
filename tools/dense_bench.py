#!/usr/bin/env python3
"""Throughput of the batched dense-trajectory kernel (lt_integrate_dense_dev) on one GPU.

usage: dense_bench.py [n_tracks] [max_points] [a] [binning]      (defaults 1048576 256 0.9 0;
       binning = lt_dense_opts.length_binning: 0 automatic, 1 predictor pass + longest-first launch, -1 caller order)

Workload: Kerr, observer at r = 50 M, viewing angles uniform in [0.01, 0.4] rad, screen angles uniform in
[0, 2 pi), the reference's solve_ivp settings (rtol 1e-8, atol 1e-10, max_step 1).  Initial states are built
on the host (metric.initial_conditions, vectorised here) and are resident in HBM when the timed region starts.
Reports tracks/s, points/s, right-hand-side evaluations/s, the record write rate against HBM, and algorithmic
float64 FLOP/s (as-written count of the reference: 218 per Kerr 8-D right-hand side, 544 per step attempt of
solve_ivp's RK45 for the stage sums, the error norm and the controller).
"""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "light-path-tracer_amd"))
import ltrace  # noqa: E402
import metrics  # noqa: E402

F_RHS, F_ATTEMPT = 218, 544
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
PEAK_F64_ISSUE = 1024 * 2.4e9 / 4      # wave-instructions per second: 1024 SIMDs, one float64 VALU instruction per 4 cycles


def roofline(key, ms_tracks):
    """k_dense_tracks against the float64 VALU issue peak, from the SQ_INSTS_VALU count committed for this workload and
    this build (profiles/valu_counts.json, written by tools/prof_dense.sh); None without a matching record."""
    rec = None
    # LT_VALU_RECORD: the record tools/prof_dense.sh has just written for this build (before it is merged into profiles/)
    for path, pick in ((os.environ.get("LT_VALU_RECORD"), lambda d: d.get(key)),
                       (os.path.join(ROOT, "profiles", "valu_counts.json"), lambda d: (d.get("workloads") or {}).get(key))):
        if not path or rec:
            continue
        try:
            with open(path) as f:
                rec = pick(json.load(f))
        except (OSError, ValueError):
            rec = None
    if not rec:
        return None
    fresh = rec.get("build_id") == ltrace.build_id()
    ms = rec["kernel_ms_in_the_profiled_run"] if ms_tracks is None else ms_tracks
    return {"bound": "valu_issue_fp64", "kernel": "k_dense_tracks", "valu_wave_insts_per_launch": rec["valu_insts"],
            "lane_utilisation": rec.get("lane_utilisation"), "kernel_ms": round(ms, 3),
            "frac": round(rec["valu_insts"] / (ms * 1e-3) / PEAK_F64_ISSUE, 4),
            "source": rec.get("source", "") + ("" if fresh else f"; STALE: measured on build {rec.get('build_id')}")}


def states(n, a, seed=3):
    rng = np.random.default_rng(seed)
    met = metrics.Kerr(1.0, a)
    al, th = rng.uniform(0.01, 0.4, n), rng.uniform(0, 2 * np.pi, n)
    # metric.initial_conditions (metrics.py:1032-1107) for theta_obs = pi/2, vectorised
    r = 50.0
    Sigma, Delta = r * r, r * r - 2 * r + a * a
    rho = r * np.sin(al) * np.sqrt(Sigma) / np.sqrt(Delta)
    xi = rho * np.sin(th)
    Theta = np.maximum((rho * np.cos(th)) ** 2, 0.0)
    p_th = np.where(np.cos(th) > 0, -1.0, 1.0) * np.sqrt(Theta)
    A = (r * r + a * a) ** 2 - a * a * Delta
    other = (-A / (Sigma * Delta) + 2 * (-2 * a * r / (Sigma * Delta)) * (-1.0) * xi + p_th ** 2 / Sigma
             + (Delta - a * a) / (Sigma * Delta) * xi ** 2)
    p_r = -np.sqrt(np.maximum(-other / (Delta / Sigma), 0.0))
    s0 = np.stack([np.zeros(n), np.full(n, r), np.full(n, np.pi / 2), np.zeros(n), np.full(n, -1.0), p_r, p_th, xi], axis=1)
    chk = np.array(met.initial_conditions(r, al[0], th[0]))
    assert np.allclose(s0[0], chk, rtol=1e-12, atol=1e-12), (s0[0], chk)
    return s0


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
    mp = int(sys.argv[2]) if len(sys.argv) > 2 else 256
    a = float(sys.argv[3]) if len(sys.argv) > 3 else 0.9
    binning = int(sys.argv[4]) if len(sys.argv) > 4 else 0
    dev = torch.device("cuda:0")
    s0 = torch.from_numpy(states(n, a)).to(dev)
    t = torch.empty((n, mp), dtype=torch.float64, device=dev)
    y = torch.empty((n, mp, 8), dtype=torch.float64, device=dev)
    cnt = torch.zeros(n, dtype=torch.int32, device=dev)
    st = torch.zeros(n, dtype=torch.int8, device=dev)
    nf = torch.zeros(n, dtype=torch.int32, device=dev)
    stream = torch.cuda.current_stream()
    o = ltrace.default_dense_opts(max_points=mp, stream=stream.cuda_stream, length_binning=binning)
    met = ltrace.Metric(ltrace.METRIC_KERR, 0, 1.0, a)
    run = lambda: ltrace.integrate_dense_dev(met, o, s0.data_ptr(), n, t.data_ptr(), y.data_ptr(), cnt.data_ptr(),
                                             st.data_ptr(), nf.data_ptr())
    run(); torch.cuda.synchronize()
    reps = 5
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    for _ in range(reps):
        run()
    e1.record(stream); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    pts = int(torch.minimum(cnt, torch.tensor(mp, device=dev, dtype=torch.int32)).sum())
    pts_full = int(cnt.sum()); evals = int(nf.sum())
    attempts = (evals - 2 * n) / 6
    flops = evals * F_RHS + attempts * F_ATTEMPT
    # the main kernel alone, by HIP events around a launch of its own (caller order: the launch is that kernel; binned: the
    # predictor and the sort run first, so the record's own time from the traced run prices the kernel)
    key = f"dense_tracks|kerr_a{a}|n{n}|mp{mp}|binning{binning}"
    roof = roofline(key, ms if binning < 0 else None)
    print(json.dumps(dict(workload=f"dense tracks Kerr a={a} r_obs=50, n={n}, max_points={mp}", length_binning=binning, ms=round(ms, 3),
                          **({"roofline": roof} if roof else {}),
                          tracks_per_s=round(n / ms * 1e3), points_per_s=round(pts / ms * 1e3),
                          points_per_track=round(pts_full / n, 1), truncated=int((cnt > mp).sum()),
                          rhs_evals_per_track=round(evals / n, 1), rejected_frac=round(1 - (pts_full - n) / attempts, 4),
                          record_write_GBps=round(pts * 72 / ms / 1e6, 1), hbm_peak_GBps=8000,
                          fp64_tflops=round(flops / ms / 1e9, 2), fp64_vector_peak_tflops=78.6,
                          endings=dict(capture=int((st == 1).sum()), escape=int((st == 2).sum()),
                                       range_end=int((st == 0).sum()), failed=int((st < 0).sum())))))


if __name__ == "__main__":
    main()
