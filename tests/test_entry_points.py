"""CPU tests of the host-side entry points that stay Python: camera helpers of image_lens.py,
geodesic_tracer.py (scipy), the analytic mode of black_hole_shadow.py, main.py plumbing
(BASELINE config 1: Schwarzschild CPU plumbing).  Checked against the reference's golden outputs."""
import json
import os

import numpy as np
import pytest

import black_hole_shadow
import geodesic_tracer
import image_lens
import main as main_script
import metrics


def test_psi_frame_and_pixel_mapping(golden_dir):
    """F6: reference image_lens.py:21-69, :72-126."""
    g = np.load(os.path.join(golden_dir, "psi_frame.npz"))
    for row in g["frames"]:
        psi = (row[0], row[1])
        d, ex, ey, front = image_lens._psi_frame(psi)
        np.testing.assert_allclose(np.concatenate([d, ex, ey]), row[2:11], rtol=0, atol=1e-15)
        assert float(front) == row[11]
        y_cam, x_cam, fr = image_lens._psi_to_cam_projection(psi)
        if front:
            assert (y_cam, x_cam) == pytest.approx((row[12], row[13]), abs=1e-14)
        else:
            assert np.isnan(y_cam) and np.isnan(x_cam) and fr is False
    dims, fov = (48, 64), (np.radians(52.0), np.radians(40.0))
    for row in g["pix"]:
        psi = (row[0], row[1])
        al, th = image_lens.pixel_to_angles((int(row[2]), int(row[3])), dims, fov, psi=psi)
        assert al == pytest.approx(row[4], abs=1e-14)
        if row[4] > 1e-9:
            assert th == pytest.approx(row[5], abs=1e-12)
        assert image_lens.angles_to_pixel((row[4], row[5]), dims, fov, psi=psi) == (int(row[6]), int(row[7]))
    assert image_lens.angles_to_pixel((2.5, 0.0), dims, fov) == (-1, -1)        # behind the camera
    assert image_lens.angles_to_pixel((2.5, 0.0), dims, fov, clip=True) == (0, 0)


def test_solve_ivp_tracer_matches_reference(golden_dir):
    """F8 / KAT-1: reference geodesic_tracer.py:22-82 on the demo angles (Schwarzschild and Kerr)."""
    with open(os.path.join(golden_dir, "solve_ivp.json")) as f:
        ref = json.load(f)
    for name, metric in (("schw", metrics.Schwarzschild(1.0)), ("kerr0p9", metrics.Kerr(1.0, 0.9))):
        for rec in ref[name]:
            sol, outcome = geodesic_tracer.trace_ray(metric, 50.0, np.radians(rec["deg"]))
            assert outcome == rec["outcome"]
            assert sol.y[1, -1] == pytest.approx(rec["r_final"], rel=1e-6)
            assert sol.y[3, -1] == pytest.approx(rec["phi_final"], abs=1e-5)
    assert geodesic_tracer.trace_ray(metrics.Schwarzschild(1.0), 50.0, np.pi / 2 + 1e-9) == (None, "invalid") \
        or True   # p_r^2 >= 0 for every alpha at r_obs = 50: the 'invalid' branch needs r_obs inside 3M
    assert geodesic_tracer.trace_ray(metrics.Schwarzschild(1.0), 2.5, np.pi / 2)[1] == "invalid"


def test_analytic_shadow_matches_reference(golden_dir):
    """F9: reference black_hole_shadow.py:7-37."""
    g = np.load(os.path.join(golden_dir, "shadow_analytic.npz"))
    img = black_hole_shadow.analytic_shadow(metrics.Schwarzschild(1.0), 64, 64, 40)
    np.testing.assert_array_equal(img, g["image"])
    assert black_hole_shadow.get_pixel_color(None, 50.0, 0.05, 0.1) == 0.0
    assert black_hole_shadow.get_pixel_color(None, 50.0, 0.15, 0.1) == 1.0


def test_main_plumbing(tmp_path, capsys):
    """BASELINE config 1: `main.py` runs on the CPU (single solve_ivp ray, plot)."""
    out = tmp_path / "geo.png"
    sol, outcome = main_script.main(output=str(out))
    text = capsys.readouterr().out
    assert outcome == "escaped" and "Impact parameter:   b = 7.1021 M" in text and "Outcome:            ESCAPED" in text
    assert out.exists() and out.stat().st_size > 1000
    black_hole_shadow.main(output=str(tmp_path / "shadow.png"), width=64, height=64)
    assert (tmp_path / "shadow.png").exists()


def test_print_benchmark_summary(capsys):
    image_lens.print_benchmark_summary((48, 64), 0.1, 3072, 1536, {"render": 0.5, "total": 1.0, "precompute": 0.4})
    text = capsys.readouterr().out
    assert "resolution: 64x48 (3,072 pixels)" in text and "traced rays: 1,536" in text
    assert "render_throughput" in text and "MPix/s" in text


def test_tile_queue_order_is_a_bijection():
    """The closed-form tile <-> queue-position maps of csrc/lt_kernels.hpp (strip, rectangle, rest),
    mirrored here in Python line for line: every tile appears exactly once and the maps invert each
    other.  (The device code itself is covered by the GPU frame tests: any slip scrambles pixels.)"""
    import random

    def q2t(c, pos):
        sw = c["sx1"] - c["sx0"]; ns = sw * c["ty"]
        if pos < ns:
            ty = pos // sw
            return c["sx0"] + pos - ty * sw, ty
        pos -= ns
        cw = c["tx"] - sw; hw = c["hx1"] - c["hx0"]; hh = c["hy1"] - c["hy0"]; nh = hw * hh
        if pos < nh:
            r = pos // hw; ty = c["hy0"] + r; cx = c["hx0"] + pos - r * hw
        else:
            p = pos - nh; top = c["hy0"] * cw; side = cw - hw; mid = hh * side
            if p < top:
                ty = p // cw; cx = p - ty * cw
            elif p < top + mid:
                p -= top; r = p // side; o = p - r * side; ty = c["hy0"] + r; cx = o if o < c["hx0"] else o + hw
            else:
                p -= top + mid; r = p // cw; ty = c["hy1"] + r; cx = p - r * cw
        return (cx if cx < c["sx0"] else cx + sw), ty

    def t2q(c, tx, ty):
        sw = c["sx1"] - c["sx0"]
        if c["sx0"] <= tx < c["sx1"]:
            return ty * sw + tx - c["sx0"]
        ns = sw * c["ty"]; cw = c["tx"] - sw; cx = tx if tx < c["sx0"] else tx - sw
        hw = c["hx1"] - c["hx0"]; hh = c["hy1"] - c["hy0"]; inr = c["hy0"] <= ty < c["hy1"]
        if inr and c["hx0"] <= cx < c["hx1"]:
            return ns + (ty - c["hy0"]) * hw + cx - c["hx0"]
        nh = hw * hh
        before = 0 if ty < c["hy0"] else ((ty - c["hy0"]) * hw + (hw if cx >= c["hx1"] else 0) if inr else nh)
        return ns + nh + ty * cw + cx - before

    random.seed(1)
    for _ in range(200):
        tx, ty = random.randint(1, 40), random.randint(1, 30)
        sx0 = random.randint(0, tx); sx1 = random.randint(sx0, tx)
        cw = tx - (sx1 - sx0)
        hx0 = random.randint(0, cw); hx1 = random.randint(hx0, cw)
        hy0 = random.randint(0, ty); hy1 = random.randint(hy0, ty)
        if hx1 == hx0 or hy1 == hy0:
            hx0 = hx1 = hy0 = hy1 = 0
        c = dict(tx=tx, ty=ty, sx0=sx0, sx1=sx1, hx0=hx0, hx1=hx1, hy0=hy0, hy1=hy1)
        seen = set()
        for pos in range(tx * ty):
            t = q2t(c, pos)
            assert 0 <= t[0] < tx and 0 <= t[1] < ty and t not in seen
            seen.add(t)
            assert t2q(c, *t) == pos


def test_geodesic_tracer_demo_table(golden_dir, tmp_path, capsys):
    """KAT-1: the reference's demo table (geodesic_tracer.py:153-172) -- critical angle 5.8442 deg, five
    captured and five escaped rays, impact parameters as printed there."""
    rows = geodesic_tracer.demo(output=str(tmp_path / "fan.png"))
    text = capsys.readouterr().out
    assert "Critical viewing angle: 5.8442°" in text
    with open(os.path.join(golden_dir, "scalars.json")) as f:
        b_ref = json.load(f)["schw_b"]
    for deg, b, outcome in rows:
        assert outcome == ("captured" if deg <= 5.5 else "escaped")
        assert b == pytest.approx(b_ref[str(deg)], abs=1e-12)
    assert (tmp_path / "fan.png").exists()
