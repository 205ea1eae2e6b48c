#!/usr/bin/env python3
"""Error statistics of the GPU batch tracers against EVERY golden per-ray fixture (tests/golden/rays_*.npz, the reference's
own outputs): what the budgets stated in tests/test_gpu_parity.py are set from.  One line per fixture and kernel."""
import glob, json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "light-path-tracer_amd"))
import ltrace
for f in sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "rays_*.npz"))):
    g = np.load(f)
    meta = json.loads(str(g["meta"]))
    n = g["alpha"].size
    runs = [("dp45", 64), ("dp45_exact", 64)] if meta["kind"] == "dp45" else [("rk4", 64), ("rk4", 32)]
    for integ, prec in runs:
        fa, w = np.full(n, np.nan), np.zeros(n, dtype=np.int64)
        st, ev = np.zeros(n, dtype=np.int8), np.zeros(n, dtype=np.uint32)
        if meta["kind"] == "schw":
            ltrace.trace_batch_schw(meta["M"], meta["r_obs"], g["alpha"], fa, w, precision=prec, out_status=st, out_rhs_evals=ev)
        else:
            ltrace.trace_batch_kerr(meta["M"], meta["a"], meta["r_obs"], g["alpha"], g["theta"], meta.get("theta_obs", np.pi / 2),
                                    max(5000.0, 6.0 * meta["r_obs"]), g["refine"], fa, w, integrator=integ, precision=prec,
                                    out_status=st, out_rhs_evals=ev)
        cls = (st == 1) == (g["status"] == 1)
        esc = cls & (st == 1)
        d = np.abs(fa[esc] - g["final_alpha"][esc])
        same = st == g["status"]
        wd = (w != g["n_half"]) & same
        print(f"{os.path.basename(f):44s} {integ:10s} f{prec}: n {n:6d} class flips {(~cls).sum():3d}  |dfa| median {np.median(d):.1e} p99 {np.quantile(d, 0.99):.1e} "
              f"max {d.max():.1e}  winding diff escaped {(wd & (st == 1)).sum():2d} captured {(wd & (st != 1)).sum():3d}  "
              f"evals differ {(ev.astype(np.int64) != g['rhs_evals']).sum():4d}  mean evals {ev.mean():.1f} vs {g['rhs_evals'].mean():.1f}")
