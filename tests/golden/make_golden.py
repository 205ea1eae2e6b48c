#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by IMPORTING the reference.

Run in the build container only (the reference lives at /root/reference and
never travels):

    MPLBACKEND=Agg PYTHONDONTWRITEBYTECODE=1 python3 tests/golden/make_golden.py [--only F3]

What is captured (inputs -> outputs, all float64 unless noted), following
SURVEY.md section 8(c):

  F1  kerr_rhs.npz        _kerr_geodesic_equations_numba   (metrics.py:221-303)
  F2  kerr_ic.npz         _kerr_initial_conditions_numba   (metrics.py:148-218)
  F3  rays_*.npz          per-ray (status, final_alpha, n_half, rhs_evals) of
                          _schwarzschild_trace_ray_numba   (metrics.py:120-145)
                          _kerr_trace_ray_numba  (DP45)    (metrics.py:419-567)
                          _kerr_trace_ray_rk4_numba        (metrics.py:570-658)
                          on n x n pinhole grids
  F4  lookup_*.npz        build_alpha_lookup / precompute_final_alpha_lookup(_2d)
                          (image_lens.py:133-280) incl. psi offsets, pins Q1/Q2
  F5  render_*.npz        render_lensed_image (image_lens.py:296-397)
  F6  psi_frame.npz       _psi_frame (image_lens.py:38-61), pixel_to_angles (:72-93)
  F7  scalars.json        alpha_crit, r_plus, impact parameters, single-ray KATs
  F8  solve_ivp.json      geodesic_tracer.trace_ray outcomes (geodesic_tracer.py:74-82)
  F10 dense_tracks.npz    geodesic_tracer.integrate_geodesic (geodesic_tracer.py:22-71): every accepted
                          solve_ivp point (t, 8-D y), nfev and outcome of 8-D initial states from
                          metric.initial_conditions (metrics.py:792-808, :1032-1107), plus the 8-D
                          right-hand sides (metrics.py:763-790, :946-1029) on random states

The fixtures are data only (numbers in, numbers out); no reference source text
is stored.  numba is absent in this container, so the reference runs through
its own pure-Python fallback (metrics.py:16-29): same lines, same float64.
"""
import argparse
import json
import os
import sys
import time
from multiprocessing import Pool

import numpy as np

REF = os.environ.get("LT_REFERENCE", "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))
os.environ.setdefault("MPLBACKEND", "Agg")
sys.dont_write_bytecode = True
sys.path.insert(0, REF)

import metrics as ref_metrics  # noqa: E402  (the reference)


# --------------------------------------------------------------------------
# helpers
# --------------------------------------------------------------------------
def pixel_grid(n_h, n_w, hfov, vfov, psi=(0.0, 0.0)):
    """alpha (f32-quantised, Q2) and theta (f64) per pixel via the reference's own
    build_alpha_lookup and pixel_to_angles-equivalent vector formula."""
    import image_lens as ref_il
    alpha32 = ref_il.build_alpha_lookup((n_h, n_w), (hfov, vfov), psi=psi)
    fx = (n_w / 2) / np.tan(hfov / 2)
    fy = (n_h / 2) / np.tan(vfov / 2)
    x_cam = (np.arange(n_w) - n_w / 2) / fx
    y_cam = (np.arange(n_h) - n_h / 2) / fy
    d, e_x, e_y, _ = ref_il._psi_frame(psi)
    den = np.sqrt(1.0 + x_cam[None, :] ** 2 + y_cam[:, None] ** 2)
    vx = x_cam[None, :] / den
    vy = y_cam[:, None] / den
    vz = 1.0 / den
    theta = np.arctan2(vx * e_x[0] + vy * e_x[1] + vz * e_x[2],
                       vx * e_y[0] + vy * e_y[1] + vz * e_y[2])
    # spot-check against the reference's scalar pixel_to_angles
    for (iy, ix) in [(0, 0), (n_h // 3, n_w // 5), (n_h - 1, n_w - 1)]:
        a_s, t_s = ref_il.pixel_to_angles((iy, ix), (n_h, n_w), (hfov, vfov), psi=psi)
        assert abs(np.float32(a_s) - alpha32[iy, ix]) <= 1e-6
        assert abs(t_s - theta[iy, ix]) <= 1e-12
    # axis-refine columns (image_lens.py:210-216)
    _, bh_x, front = ref_il._psi_to_cam_projection(psi)
    if front:
        x_rel = x_cam - bh_x
        m = max(float(np.max(np.abs(x_rel))), 1e-12)
        cols = np.abs(x_rel) <= ref_il.Y_AXIS_REFINE_FRAC * m
    else:
        cols = np.zeros(n_w, dtype=bool)
    return alpha32, theta, cols


_COUNTER = [0]


def _install_counters():
    if getattr(ref_metrics, "_lt_counted", False):
        return
    orig_k = ref_metrics._kerr_geodesic_equations_numba
    orig_s = ref_metrics._schwarzschild_orbit_rhs_numba

    def k(*a):
        _COUNTER[0] += 1
        return orig_k(*a)

    def s(*a):
        _COUNTER[0] += 1
        return orig_s(*a)

    ref_metrics._kerr_geodesic_equations_numba = k
    ref_metrics._schwarzschild_orbit_rhs_numba = s
    ref_metrics._lt_counted = True


def _trace_one(job):
    kind, M, a, r_obs, alpha, theta, refine, theta_obs = job
    _install_counters()
    _COUNTER[0] = 0
    if kind == "schw":
        s, fa, nh = ref_metrics._schwarzschild_trace_ray_numba(
            M, 2.0 * M, r_obs, alpha, 50.0, 0.05)
    else:
        r_plus = M + np.sqrt(M * M - a * a)
        lam = max(5000.0, 6.0 * r_obs)
        fn = (ref_metrics._kerr_trace_ray_numba if kind == "dp45"
              else ref_metrics._kerr_trace_ray_rk4_numba)
        s, fa, nh = fn(M, a, r_plus, r_obs, alpha, theta, theta_obs, lam, 1.0, bool(refine))
    return int(s), float(fa), int(nh), int(_COUNTER[0])


def trace_grid(pool, kind, M, a, r_obs, n, refine_mode, fov_deg=40.0, theta_obs=np.pi / 2):
    fov = np.radians(fov_deg)
    alpha32, theta, cols = pixel_grid(n, n, fov, fov)
    alpha = alpha32.astype(np.float64).ravel()
    th = theta.ravel()
    if refine_mode == "cols":
        refine = np.broadcast_to(cols[None, :], (n, n)).ravel()
    else:
        refine = np.zeros(n * n, dtype=bool)
    jobs = [(kind, M, a, r_obs, float(alpha[i]), float(th[i]), bool(refine[i]), float(theta_obs))
            for i in range(n * n)]
    t0 = time.time()
    out = pool.map(_trace_one, jobs, chunksize=64)
    dt = time.time() - t0
    status = np.array([o[0] for o in out], dtype=np.int8)
    fa = np.array([o[1] for o in out], dtype=np.float64)
    nh = np.array([o[2] for o in out], dtype=np.int32)
    ev = np.array([o[3] for o in out], dtype=np.int32)
    print(f"  {kind} a={a} r_obs={r_obs} n={n} refine={refine_mode}: {dt:.1f}s "
          f"({n*n/dt:.0f} rays/s), mean evals {ev.mean():.1f}, max {ev.max()}")
    return dict(kind=kind, M=M, a=a, r_obs=r_obs, n=n, fov_deg=fov_deg, theta_obs=float(theta_obs),
                refine_mode=refine_mode, alpha=alpha, theta=th,
                refine=refine.astype(np.uint8), status=status, final_alpha=fa,
                n_half=nh, rhs_evals=ev)


OUT_DIR = HERE      # --verify writes into a temporary directory instead and compares


def save(name, **arrs):
    path = os.path.join(OUT_DIR, name)
    np.savez_compressed(path, **arrs)
    print(f"  wrote {name} ({os.path.getsize(path)/1024:.0f} KiB)")


# --------------------------------------------------------------------------
# fixtures
# --------------------------------------------------------------------------
def f1_rhs():
    rng = np.random.default_rng(20260101)
    recs_in, recs_out = [], []
    for a in (0.0, 0.5, 0.9, 0.99):
        M = 1.0
        r_plus = M + np.sqrt(M * M - a * a)
        for _ in range(64):
            r = rng.uniform(1.02 * r_plus, 200.0) if rng.random() < 0.7 else rng.uniform(1.02 * r_plus, 6.0)
            th = rng.uniform(1e-3, np.pi - 1e-3)
            st = np.array([r, th, rng.uniform(-6, 6), rng.uniform(-10, 10), rng.uniform(-10, 10)])
            p_t = -1.0
            p_phi = rng.uniform(-10, 10)
            out = np.empty(5)
            ref_metrics._kerr_geodesic_equations_numba(st, p_t, p_phi, M, a, r_plus, out)
            recs_in.append(np.concatenate([st, [p_t, p_phi, M, a, r_plus]]))
            recs_out.append(out.copy())
        # inside the 1.001 r_plus cut and on the pole floor
        for st in ([1.0005 * r_plus, 1.0, 0.0, -1.0, 0.5], [5.0, 0.0, 0.0, -0.3, 0.2],
                   [5.0, np.pi, 1.0, 0.3, -0.2]):
            st = np.array(st, dtype=np.float64)
            out = np.empty(5)
            ref_metrics._kerr_geodesic_equations_numba(st, -1.0, 2.0, M, a, r_plus, out)
            recs_in.append(np.concatenate([st, [-1.0, 2.0, M, a, r_plus]]))
            recs_out.append(out.copy())
    save("kerr_rhs.npz", inputs=np.array(recs_in), outputs=np.array(recs_out))


def f2_ic():
    ins, outs = [], []
    for a in (0.0, 0.5, 0.9, 0.99):
        for r_obs in (50.0, 100.0, 8.0):
            for theta_obs in (np.pi / 2, 1.0):
                for alpha in (0.0, 1e-3, 0.05, 0.15, 0.3, 1.2, 2.0):
                    for th in (-np.pi / 2, -0.7, 0.0, 0.7, np.pi / 2, 3.0):
                        ok, st, p_t, p_phi = ref_metrics._kerr_initial_conditions_numba(
                            1.0, a, r_obs, alpha, th, theta_obs)
                        ins.append([1.0, a, r_obs, alpha, th, theta_obs])
                        outs.append([float(ok)] + list(st) + [p_t, p_phi])
    save("kerr_ic.npz", inputs=np.array(ins), outputs=np.array(outs))


def f3_rays(sizes, skip_existing=False):
    with Pool(8) as pool:
        for case in sizes:
            kind, a, r_obs, n, mode = case[:5]
            theta_obs = case[5] if len(case) > 5 else np.pi / 2      # observer inclination (metrics.py:148-218 takes any)
            name = f"rays_{kind}_a{str(a).replace('.', 'p').replace('-', 'm')}_r{int(r_obs)}_n{n}_{mode}"
            if theta_obs != np.pi / 2:
                name += "_th" + f"{theta_obs:g}".replace(".", "p")
            name += ".npz"
            if skip_existing and os.path.exists(os.path.join(HERE, name)):
                continue
            g = trace_grid(pool, kind, 1.0, a, r_obs, n, mode, theta_obs=theta_obs)
            meta = {k: g[k] for k in ("kind", "M", "a", "r_obs", "n", "fov_deg", "refine_mode", "theta_obs")}
            save(name, meta=json.dumps(meta),
                 alpha=g["alpha"], theta=g["theta"], refine=g["refine"],
                 status=g["status"], final_alpha=g["final_alpha"],
                 n_half=g["n_half"], rhs_evals=g["rhs_evals"])


def f4_lookup():
    import image_lens as ref_il
    cases = [
        ("4x6_a0", (4, 6), 0.0, 50.0, (0.0, 0.0)),
        ("4x6_a0p9", (4, 6), 0.9, 50.0, (0.0, 0.0)),
        ("48x64_a0", (48, 64), 0.0, 100.0, (0.0, 0.0)),
        ("48x64_a0p9", (48, 64), 0.9, 100.0, (0.0, 0.0)),
        ("48x64_a0_psi", (48, 64), 0.0, 100.0, (0.1, -0.2)),
        ("48x64_a0p9_psi", (48, 64), 0.9, 100.0, (0.1, -0.2)),
        ("33x40_a0p9", (33, 40), 0.9, 50.0, (0.0, 0.0)),   # odd height: Q1 with (H+1)//2 rows
    ]
    for name, (h, w), a, r_obs, psi in cases:
        metric = ref_metrics.Kerr(1.0, a) if a != 0 else ref_metrics.Schwarzschild(1.0)
        vfov = np.radians(40.0)
        hfov = 2 * np.arctan(np.tan(vfov / 2) * w / h)
        fov = (hfov, vfov)
        ac = metric.alpha_crit(r_obs)
        al = ref_il.build_alpha_lookup((h, w), fov, psi=psi)
        if metric.is_spherically_symmetric:
            fa, wd, total, traced = ref_il.precompute_final_alpha_lookup(al, ac, r_obs, metric)
        else:
            fa, wd, total, traced = ref_il.precompute_final_alpha_lookup_2d(
                al, fov, ac, r_obs, metric, psi=psi)
        # synthetic background, SURVEY F5 pattern
        yy, xx = np.mgrid[0:h, 0:w]
        bg = np.stack([(4 * xx) % 256, (5 * yy) % 256, 255 * ((xx // 8 + yy // 8) % 2)],
                      axis=-1).astype(np.float32) / 255.0
        img = ref_il.render_lensed_image(bg, al, fa, wd, ac, fov, False, psi=psi)
        img_wrap = ref_il.render_lensed_image(bg, al, fa, wd, ac, fov, True, psi=psi)
        gray = ref_il.render_lensed_image(bg[..., 1].copy(), al, fa, wd, ac, fov, False, psi=psi)
        save(f"lookup_{name}.npz",
             meta=json.dumps(dict(h=h, w=w, a=a, M=1.0, r_obs=r_obs, psi=list(psi),
                                  hfov=hfov, vfov=vfov, alpha_crit=float(ac),
                                  total=int(total), traced=int(traced))),
             alpha_lookup=al, final_alpha=fa, winding=wd, background=bg,
             lensed=img, lensed_wrap=img_wrap, lensed_gray=gray)


def f6_psi():
    import image_lens as ref_il
    psis = [(0.0, 0.0), (0.1, -0.2), (-0.3, 0.4), (0.0, 1.0), (1.2, 0.0),
            (np.pi / 2, 0.0), (0.0, np.pi / 2), (0.2, 2.5)]
    rows = []
    for psi in psis:
        d, ex, ey, front = ref_il._psi_frame(psi)
        y_cam, x_cam, fr2 = ref_il._psi_to_cam_projection(psi)
        rows.append(list(psi) + list(d) + list(ex) + list(ey) + [float(front), y_cam, x_cam])
    # pixel_to_angles / angles_to_pixel samples
    pa = []
    dims = (48, 64)
    fov = (np.radians(52.0), np.radians(40.0))
    for psi in psis[:4]:
        for (iy, ix) in [(0, 0), (10, 50), (24, 32), (47, 63), (24, 33)]:
            al, th = ref_il.pixel_to_angles((iy, ix), dims, fov, psi=psi)
            py, px = ref_il.angles_to_pixel((al, th), dims, fov, psi=psi)
            pa.append(list(psi) + [iy, ix, al, th, py, px])
    save("psi_frame.npz", frames=np.array(rows, dtype=np.float64), pix=np.array(pa, dtype=np.float64))


def f7_scalars():
    out = {}
    S = ref_metrics.Schwarzschild(1.0)
    out["schw_alpha_crit"] = {str(r): float(S.alpha_crit(r)) for r in (8.0, 50.0, 100.0)}
    out["schw_capture_radius"] = float(S.capture_radius())
    out["schw_b"] = {str(d): float(S.viewing_angle_to_impact_parameter(np.radians(d), 50.0))
                     for d in (0, 2, 4, 5, 5.5, 5.97, 6.5, 8, 10, 15)}
    out["schw_trace_ray"] = {}
    for al in (0.0, 0.05, 0.1019, 0.1021, 0.103, 0.15, 0.3, 1.0, 2.5):
        fa, nh, oc = S.trace_ray(50.0, al)
        out["schw_trace_ray"][repr(al)] = [None if np.isnan(fa) else float(fa), int(nh), oc]
    out["kerr"] = {}
    for a in (0.0, 0.5, 0.9, 0.99, -0.9, 1.0):
        K = ref_metrics.Kerr(1.0, a)
        rec = dict(r_plus=float(K.r_plus), capture_radius=float(K.capture_radius()),
                   alpha_crit={str(r): float(K.alpha_crit(r)) for r in (50.0, 100.0)},
                   alpha_crit_incl=float(K.alpha_crit(50.0, 1.0)),
                   b=float(K.viewing_angle_to_impact_parameter(0.15, 50.0)),
                   photon_r=[float(x) for x in K._unstable_photon_r()])
        if a != 0:
            rec["crit"] = [[float(x), float(y)] for x, y in K._critical_impact_params()]
        rays = {}
        if abs(a) < 1.0:
            for (al, th, rf) in [(0.15, 0.7, False), (0.09, np.pi / 2, False), (0.09, -np.pi / 2, False),
                                 (0.2, 0.0, True), (0.136, 1.0, False), (0.5, -2.0, False)]:
                lam = max(5000.0, 6.0 * 50.0)
                dp = ref_metrics._kerr_trace_ray_numba(1.0, a, K.r_plus, 50.0, al, th, np.pi / 2, lam, 1.0, rf)
                rk = ref_metrics._kerr_trace_ray_rk4_numba(1.0, a, K.r_plus, 50.0, al, th, np.pi / 2, lam, 1.0, rf)
                cv = lambda t: [int(t[0]), None if np.isnan(t[1]) else float(t[1]), int(t[2])]
                rays[f"{al!r},{th!r},{int(rf)}"] = dict(dp45=cv(dp), rk4=cv(rk))
                fa, nh, oc = K.trace_ray(50.0, al, th, axis_refine=rf)
                rays[f"{al!r},{th!r},{int(rf)}"]["trace_ray"] = [None if np.isnan(fa) else float(fa), int(nh), oc]
        rec["rays"] = rays
        out["kerr"][repr(a)] = rec
    try:
        ref_metrics.Kerr(1.0, 1.5)
        out["kerr_bad_spin"] = "no error"
    except ValueError as e:
        out["kerr_bad_spin"] = "ValueError"
    # 8-D plugin surface
    K = ref_metrics.Kerr(1.0, 0.9)
    s0 = K.initial_conditions(50.0, 0.15, 0.7)
    out["kerr_ic8"] = [float(x) for x in s0]
    out["kerr_rhs8"] = [float(x) for x in K.geodesic_equations(0.0, s0)]
    s0 = S.initial_conditions(50.0, 0.15)
    out["schw_ic8"] = [float(x) for x in s0]
    out["schw_rhs8"] = [float(x) for x in S.geodesic_equations(0.0, s0)]
    out["schw_ic8_none"] = S.initial_conditions(50.0, np.pi / 2 + 1e-9) is None or "not none"
    with open(os.path.join(OUT_DIR, "scalars.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print("  wrote scalars.json")


def f8_ivp():
    import geodesic_tracer as ref_gt
    out = {}
    for name, metric in (("schw", ref_metrics.Schwarzschild(1.0)), ("kerr0p9", ref_metrics.Kerr(1.0, 0.9))):
        rows = []
        for deg in (0, 2, 4, 5, 5.5, 5.97, 6.5, 8, 10, 15):
            sol, oc = ref_gt.trace_ray(metric, 50.0, np.radians(deg))
            rows.append(dict(deg=deg, outcome=oc, r_final=float(sol.y[1, -1]),
                             phi_final=float(sol.y[3, -1]), nfev=int(sol.nfev)))
        out[name] = rows
    with open(os.path.join(OUT_DIR, "solve_ivp.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print("  wrote solve_ivp.json")


def f10_dense():
    import geodesic_tracer as ref_gt
    S, K9, K99 = ref_metrics.Schwarzschild(1.0), ref_metrics.Kerr(1.0, 0.9), ref_metrics.Kerr(1.0, 0.99)
    jobs = []  # (metric id, M, a, r_obs, alpha, theta, kwargs)
    for deg in (0, 2, 4, 5, 5.5, 5.97, 6.5, 8, 10, 15):
        jobs.append((0, S, 50.0, np.radians(deg), 0.0, {}))
    for deg in (0, 3, 6, 7.5, 7.7, 8.5, 12):
        for th in (0.0, 0.9, np.pi / 2, 2.2, 4.0, 3 * np.pi / 2):
            jobs.append((1, K9, 50.0, np.radians(deg), th, {}))
    for deg, th in ((6.8, 1.3), (7.9, 4.71), (9.0, 0.2)):
        jobs.append((2, K99, 50.0, np.radians(deg), th, {}))
    # non-default stops: affine range ends first; caller-chosen radii; a far observer
    jobs.append((0, S, 50.0, np.radians(8.0), 0.0, dict(lambda_max=30.0)))
    jobs.append((1, K9, 50.0, np.radians(9.0), 1.0, dict(lambda_max=75.5)))
    jobs.append((1, K9, 50.0, np.radians(10.0), 2.0, dict(r_stop_inner=6.0, r_stop_outer=60.0)))
    jobs.append((0, S, 200.0, np.radians(1.6), 0.0, {}))
    mid, Ma, state0, stops, offs, ts, ys, nfev, outcome, status = [], [], [], [], [0], [], [], [], [], []
    for m_id, metric, r_obs, alpha, theta, kw in jobs:
        s0 = metric.initial_conditions(r_obs, alpha, theta)
        assert s0 is not None
        sol, oc = ref_gt.integrate_geodesic(metric, s0, **kw)
        mid.append(m_id); Ma.append((metric.M, getattr(metric, "a", 0.0)))
        state0.append(s0)
        stops.append((kw.get("lambda_max", 1000.0), kw.get("r_stop_inner", metric.capture_radius()),
                      kw.get("r_stop_outer", 2.0 * s0[1])))
        ts.append(sol.t); ys.append(sol.y); offs.append(offs[-1] + len(sol.t))
        nfev.append(sol.nfev); outcome.append(1 if oc == "escaped" else -1); status.append(sol.status)
    # 8-D right-hand sides on random (off-shell) states
    rng = np.random.default_rng(10)
    rs, ro = [], []
    for metric in (S, K9, K99):
        for _ in range(64):
            r_lo = 1.02 * (metric.R_S if metric is S else metric.r_plus)
            st = [rng.uniform(-5, 5), rng.uniform(r_lo, 120.0), rng.uniform(1e-3, np.pi - 1e-3), rng.uniform(-7, 7),
                  rng.uniform(-1.5, -0.5), rng.uniform(-2, 2), rng.uniform(-8, 8), rng.uniform(-8, 8)]
            rs.append(st); ro.append(np.asarray(metric.geodesic_equations(0.0, st), dtype=np.float64))
    save("dense_tracks.npz", metric_id=np.array(mid, np.int32), M_a=np.array(Ma), state0=np.array(state0),
         stops=np.array(stops), offsets=np.array(offs, np.int64), t=np.concatenate(ts),
         y=np.concatenate(ys, axis=1), nfev=np.array(nfev, np.int64), outcome=np.array(outcome, np.int8),
         ivp_status=np.array(status, np.int8), rhs_M_a=np.array([(1.0, 0.0)] * 64 + [(1.0, 0.9)] * 64 + [(1.0, 0.99)] * 64),
         rhs_state=np.array(rs), rhs_out=np.array(ro))


def f9_shadow():
    """black_hole_shadow.py analytic image (its main() only plots; restate the
    loop over the reference's own helper functions at a small size)."""
    import black_hole_shadow as ref_bs
    S = ref_metrics.Schwarzschild(1.0)
    n = 64
    fov = np.radians(40)
    ac = S.alpha_crit(50.0)
    img = np.zeros((n, n))
    for j in range(n):
        for i in range(n):
            ax = ref_bs.pixel_to_viewing_angle(i, n, fov)
            ay = ref_bs.pixel_to_viewing_angle(j, n, fov)
            img[i, j] = ref_bs.get_pixel_color(S, 50.0, np.arccos(np.cos(ax) * np.cos(ay)), ac)
    save("shadow_analytic.npz", image=img, alpha_crit=np.array(ac))


RAY_SETS = [
    # kind, a, r_obs, n, refine
    ("schw", 0.0, 50.0, 96, "off"),
    ("schw", 0.0, 100.0, 64, "off"),
    ("rk4", 0.9, 50.0, 64, "off"),
    ("rk4", 0.9, 50.0, 64, "cols"),
    ("dp45", 0.9, 50.0, 64, "off"),
    ("dp45", 0.9, 50.0, 64, "cols"),
    ("rk4", 0.99, 50.0, 64, "cols"),
    ("dp45", 0.99, 50.0, 64, "cols"),
    ("rk4", 0.9, 100.0, 64, "cols"),
    ("dp45", 0.9, 100.0, 64, "cols"),
    ("rk4", 0.5, 50.0, 48, "cols"),
    ("rk4", 0.9, 50.0, 128, "cols"),
    ("dp45", 0.9, 50.0, 128, "cols"),
    ("rk4", 0.9, 50.0, 256, "cols"),
    # round 3: the classes the GPU frame tests exercise beyond a in {0, .5, .9, .99} and r_obs in {50, 100}
    # (tests/test_gpu_parity.py FRAMES): spin pointing the other way, extremal spin, slow spin seen from close by,
    # observers at 12 M and 200 M, an inclined observer (the reference's tracers take any theta_obs, metrics.py:148-218)
    ("rk4", -0.7, 50.0, 48, "cols"),
    ("dp45", -0.7, 50.0, 48, "cols"),
    ("rk4", 1.0, 50.0, 48, "cols"),
    ("dp45", 1.0, 50.0, 48, "cols"),
    ("rk4", 0.3, 30.0, 48, "cols"),
    ("schw", 0.0, 12.0, 48, "off"),
    ("rk4", 0.9, 12.0, 48, "cols"),
    ("dp45", 0.9, 12.0, 48, "cols"),
    ("schw", 0.0, 200.0, 48, "off"),
    ("rk4", 0.9, 200.0, 48, "cols"),
    ("dp45", 0.9, 200.0, 48, "cols"),
    ("rk4", 0.9, 50.0, 48, "cols", 1.0),
    ("dp45", 0.9, 50.0, 48, "cols", 1.0),
    ("rk4", 0.9, 300.0, 32, "cols"),
]


def verify(todo):
    """The committed fixtures are what the reference under LT_REFERENCE computes today: regenerate, compare, write nothing
    into the tree.  (tests/test_oracle_golden.py runs this where the reference is present.)"""
    global OUT_DIR
    import tempfile
    makers = {"F1": f1_rhs, "F2": f2_ic, "F6": f6_psi, "F7": f7_scalars, "F8": f8_ivp, "F9": f9_shadow, "F10": f10_dense, "F4": f4_lookup}
    bad = []
    with tempfile.TemporaryDirectory() as tmp:
        OUT_DIR = tmp
        for k in todo:
            makers[k]()
        for name in sorted(os.listdir(tmp)):
            new, old = os.path.join(tmp, name), os.path.join(HERE, name)
            if not os.path.exists(old):
                bad.append(f"{name}: not committed")
            elif name.endswith(".npz"):
                with np.load(new, allow_pickle=False) as a, np.load(old, allow_pickle=False) as b:
                    if sorted(a.files) != sorted(b.files):
                        bad.append(f"{name}: arrays {sorted(a.files)} != {sorted(b.files)}")
                    else:
                        bad += [f"{name}[{f}] differs" for f in a.files
                                if not (a[f].shape == b[f].shape and (np.array_equal(a[f], b[f], equal_nan=True) if a[f].dtype.kind == "f"
                                                                        else np.array_equal(a[f], b[f])))]
            else:
                with open(new) as fa, open(old) as fb:
                    if json.load(fa) != json.load(fb):
                        bad.append(f"{name}: JSON differs")
    OUT_DIR = HERE
    print("verify:", "ok, " + ", ".join(todo) + " reproduce" if not bad else "; ".join(bad))
    return 1 if bad else 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default="")
    ap.add_argument("--new-only", action="store_true", help="F3: only the ray sets whose file does not exist yet")
    ap.add_argument("--verify", action="store_true", help="regenerate the quick sets (default F1,F2,F6,F7, or --only) into a temporary "
                                                          "directory and compare them with the committed files, value for value")
    args = ap.parse_args()
    if args.verify:
        return verify(args.only.split(",") if args.only else ["F1", "F2", "F6", "F7"])
    todo = args.only.split(",") if args.only else ["F1", "F2", "F3", "F4", "F6", "F7", "F8", "F9", "F10"]
    t0 = time.time()
    if "F1" in todo:
        print("F1"); f1_rhs()
    if "F2" in todo:
        print("F2"); f2_ic()
    if "F6" in todo:
        print("F6"); f6_psi()
    if "F7" in todo:
        print("F7"); f7_scalars()
    if "F8" in todo:
        print("F8"); f8_ivp()
    if "F9" in todo:
        print("F9"); f9_shadow()
    if "F10" in todo:
        print("F10"); f10_dense()
    if "F4" in todo:
        print("F4"); f4_lookup()
    if "F3" in todo:
        print("F3"); f3_rays(RAY_SETS, skip_existing=args.new_only)
    print(f"done in {time.time()-t0:.0f}s")


if __name__ == "__main__":
    sys.exit(main())
