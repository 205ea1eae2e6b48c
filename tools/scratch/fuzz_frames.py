#!/usr/bin/env python3
"""Random frames through lt_render against the oracle: cameras, metrics, observers, sizes and options drawn at random
(seeded), every integrator and schedule.  Not a test of tolerances (tests/test_gpu_parity.py states those on chosen frames)
but a search for crashes, inconsistent counters, a colouring that is not bit-exact on the GPU's own lookup, and outliers.

usage: fuzz_frames.py [n_cases] [seed]      (GPU box; uses oracle/ as the checker, like the tests)"""
import os
import sys
import time

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, os.path.join(ROOT, "light-path-tracer_amd"))
sys.path.insert(0, ROOT)
import ltrace  # noqa: E402
from oracle import oracle  # noqa: E402


def main():
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    worst = dict(flip_frac=(0, None), p99=(0, None))
    t0 = time.time()
    bad = 0
    for case in range(n_cases):
        kind = "kerr" if rng.random() < 0.75 else "schwarzschild"
        a = 0.0 if kind == "schwarzschild" else float(rng.choice([rng.uniform(-1, 1), 0.9, 0.99, -0.5, 1.0, 0.0]))
        r_plus = 1 + np.sqrt(max(0.0, 1 - a * a))
        r_obs = float(rng.choice([rng.uniform(3.0 * r_plus, 40), rng.uniform(40, 400), 50.0, 100.0]))
        W, H = int(rng.integers(1, 180)), int(rng.integers(1, 140))
        vfov = np.radians(rng.uniform(5, 100))
        hfov = 2 * np.arctan(np.tan(vfov / 2) * W / H)
        psi = (0.0, 0.0) if rng.random() < 0.5 else (float(rng.normal(0, 0.3)), float(rng.normal(0, 0.8)))
        tb = bool(rng.random() < 0.2) and kind == "kerr"
        integ, prec = [("rk4", 32), ("rk4", 64), ("dp45", 64), ("dp45_exact", 64)][int(rng.integers(0, 4))] if kind == "kerr" else ("rk4", int(rng.choice([32, 64])))
        sched = "queue" if rng.random() < 0.3 else "direct"
        loop = bool(rng.random() < 0.2)
        gray = bool(rng.random() < 0.2)
        parts = 1 if tb else int(rng.choice([1, 1, 2, 3]))      # (the mirror is a whole-frame option)
        theta_obs = np.pi / 2 if (tb or rng.random() < 0.6) else float(rng.uniform(0.15, np.pi - 0.15))
        desc = dict(case=case, kind=kind, a=round(a, 4), r_obs=round(r_obs, 2), W=W, H=H, vfov=round(float(np.degrees(vfov)), 1), psi=tuple(round(p, 3) for p in psi),
                    tb=tb, theta_obs=round(theta_obs, 3), integ=integ, prec=prec, sched=sched, loop=loop, gray=gray, parts=parts)
        try:
            cam = ltrace.Camera(W, H, hfov, vfov, psi[0], psi[1], r_obs, theta_obs)
            met = ltrace.Metric(0 if kind == "schwarzschild" else 1, 0, 1.0, a)
            bg = (rng.integers(0, 256, size=(H, W) if gray else (H, W, 3), dtype=np.uint8).astype(np.float32) / 255.0)
            acc = None
            rays = 0
            for p in range(parts):
                o = ltrace.default_opts(integrator=integ, precision=prec, schedule=sched, tb_symmetry=int(tb), loop_around=int(loop),
                                        n_parts=parts, part=p, row_block=int(rng.choice([8, 16, 5])) if parts > 1 and p == 0 else 0)
                if p == 0:
                    rb = o.row_block
                o.row_block = rb
                out = ltrace.render(cam, met, o, background=bg)
                if acc is None:
                    acc = {k: np.zeros((H,) + v.shape[1:], dtype=v.dtype) for k, v in out.items() if k != "stats"}
                rows = ltrace.global_rows(H, rb or 16, parts, p)
                for k in acc:
                    acc[k][rows] = out[k]
                rays += out["stats"]["rays"]
                st = out["stats"]
                assert st["escaped"] + st["captured"] + st["invalid"] == st["rays"], ("counters", st)
            ref = oracle.lookup(kind, 1.0, a, r_obs, H, W, hfov, vfov, psi=psi, theta_obs=theta_obs, integrator="rk4" if integ == "rk4" else "dp45", tb_symmetry=tb)
            assert rays == ref["traced"] or tb, ("rays", rays, ref["traced"])
            esc_g, esc_r = acc["status"] == 1, ref["status"] == 1
            flips = int((esc_g != esc_r).sum())
            both = esc_g & esc_r
            d = np.abs(acc["fa"][both].astype(np.float64) - ref["fa"][both]) if both.any() else np.zeros(1)
            assert np.array_equal(np.isnan(acc["fa"]), ~esc_g), "NaN pattern"
            img = oracle.render(bg, acc["fa"], acc["winding"], hfov, vfov, psi=psi, loop_around=loop)
            assert np.array_equal(acc["rgb"], img), "colouring of the GPU's own lookup is not bit-exact"
            assert np.array_equal(acc["rgba"], oracle.rgba8(img)), "rgba8"
            ff = flips / (W * H)
            p99 = float(np.quantile(d, 0.99))
            if ff > worst["flip_frac"][0]:
                worst["flip_frac"] = (ff, desc)
            if p99 > worst["p99"][0]:
                worst["p99"] = (p99, desc)
            lim_f, lim_p = (3e-3, 1e-3) if prec == 32 else (1e-3, 1e-5 if integ != "dp45" else 1e-3)
            if ff > lim_f + 2.0 / (W * H) or p99 > lim_p:
                print("OUTLIER", desc, "flips", flips, "p99", p99, flush=True)
        except Exception as e:       # noqa: BLE001 (report and go on: this is a search)
            bad += 1
            print("FAIL", desc, type(e).__name__, str(e)[:300], flush=True)
    print(f"{n_cases} cases, {bad} failures, {time.time() - t0:.0f} s; worst flip fraction {worst['flip_frac']}; worst p99 |d final_alpha| {worst['p99']}")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
