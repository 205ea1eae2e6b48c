"""GPU tests of the image pipeline surface (image_lens.py / black_hole_shadow.py of the product):
each reference stage has a GPU twin, the twins agree with the reference's golden outputs, and the
fused path equals the staged path bit for bit."""
import json
import os

import numpy as np
import pytest

import black_hole_shadow
import image_lens
import ltrace
import metrics
from oracle import oracle

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")
LOOKUPS = ["4x6_a0", "4x6_a0p9", "48x64_a0", "48x64_a0p9", "48x64_a0_psi", "48x64_a0p9_psi", "33x40_a0p9"]


def _g(name):
    g = np.load(os.path.join(GOLD, f"lookup_{name}.npz"))
    return g, json.loads(str(g["meta"]))


@pytest.mark.parametrize("name", LOOKUPS)
def test_stage1_alpha_lookup_is_bit_exact(name):
    """build_alpha_lookup (reference image_lens.py:133-152): float32, bit for bit; theta to 1e-12."""
    g, m = _g(name)
    fov, psi = (m["hfov"], m["vfov"]), tuple(m["psi"])
    al = image_lens.build_alpha_lookup((m["h"], m["w"]), fov, psi=psi)
    assert al.dtype == np.float32
    np.testing.assert_array_equal(al, g["alpha_lookup"])
    cam = ltrace.Camera(m["w"], m["h"], fov[0], fov[1], psi[0], psi[1], 50.0, np.pi / 2)
    _, th, cols = ltrace.pixel_angles(cam)
    _, th_o, cols_o = oracle.pixel_angles(m["h"], m["w"], fov[0], fov[1], psi=psi)
    far = al > 1e-6
    assert np.max(np.abs(th[far] - th_o[far])) < 1e-12
    np.testing.assert_array_equal(cols, cols_o)


@pytest.mark.parametrize("name", LOOKUPS)
def test_stage3_render_matches_reference_bit_for_bit(name):
    """render_lensed_image (reference image_lens.py:296-397) fed with the REFERENCE's lookups."""
    g, m = _g(name)
    fov, psi = (m["hfov"], m["vfov"]), tuple(m["psi"])
    al = g["alpha_lookup"]
    img = image_lens.render_lensed_image(g["background"], al, g["final_alpha"], g["winding"], m["alpha_crit"], fov, psi=psi)
    np.testing.assert_array_equal(img, g["lensed"])
    wrap = image_lens.render_lensed_image(g["background"], al, g["final_alpha"], g["winding"], m["alpha_crit"], fov,
                                          render_loop_around=True, psi=psi)
    np.testing.assert_array_equal(wrap, g["lensed_wrap"])
    gray = image_lens.render_lensed_image(g["background"][..., 1].copy(), al, g["final_alpha"], g["winding"],
                                          m["alpha_crit"], fov, psi=psi)
    np.testing.assert_array_equal(gray, g["lensed_gray"])
    # winding lookup omitted -> colour index 0 (image_lens.py:326-327)
    none_w = image_lens.render_lensed_image(g["background"], al, g["final_alpha"], None, m["alpha_crit"], fov, psi=psi)
    exp = oracle.render(g["background"], g["final_alpha"], None, fov[0], fov[1], psi=psi)
    np.testing.assert_array_equal(none_w, exp)


@pytest.mark.parametrize("name", ["48x64_a0", "48x64_a0_psi", "4x6_a0"])
def test_stage2_schwarzschild_lookup_matches_reference(name):
    """precompute_final_alpha_lookup (reference image_lens.py:155-178), float64 and float32 kernels."""
    g, m = _g(name)
    for precision, tol in ((64, 1e-6), (32, 2e-4)):
        S = metrics.Schwarzschild(1.0, precision=precision)
        fa, wd, total, traced = image_lens.precompute_final_alpha_lookup(g["alpha_lookup"], m["alpha_crit"], m["r_obs"], S)
        assert (total, traced) == (m["total"], m["traced"]) and fa.dtype == np.float32 and wd.dtype == np.uint16
        assert np.array_equal(np.isnan(fa), np.isnan(g["final_alpha"]))
        assert np.nanmax(np.abs(fa - g["final_alpha"])) <= tol
        assert (wd != g["winding"]).sum() == 0


def test_stage2_kerr_lookup_reproduces_tb_mirror_quirk():
    """precompute_final_alpha_lookup_2d (reference image_lens.py:185-280) incl. quirk Q1, against the
    oracle running the same integrator (RK4, float64) on an odd-height frame."""
    g, m = _g("33x40_a0p9")
    fov = (m["hfov"], m["vfov"])
    K = metrics.Kerr(1.0, 0.9, integrator="rk4", precision=64)
    fa, wd, total, traced = image_lens.precompute_final_alpha_lookup_2d(g["alpha_lookup"], fov, m["alpha_crit"], m["r_obs"], K)
    assert (total, traced) == (m["total"], m["traced"]) == (33 * 40, 17 * 40)
    ref = oracle.lookup("kerr", 1.0, 0.9, m["r_obs"], 33, 40, fov[0], fov[1], integrator="rk4", tb_symmetry=True)
    assert np.array_equal(np.isnan(fa), np.isnan(ref["fa"]))
    assert np.nanmax(np.abs(fa - ref["fa"])) < 1e-6
    np.testing.assert_array_equal(wd, ref["winding"])
    np.testing.assert_array_equal(fa[33 - 16:], fa[:16][::-1])          # row j copied to row H-1-j
    # psi_y != 0 switches the mirror off (image_lens.py:218-219)
    fa2, _, _, traced2 = image_lens.precompute_final_alpha_lookup_2d(g["alpha_lookup"], fov, m["alpha_crit"], m["r_obs"], K,
                                                                     psi=(0.1, 0.0))
    assert traced2 == 33 * 40


@pytest.mark.parametrize("a", [0.0, 0.9])
def test_fused_path_equals_staged_path(a):
    """lt_render == build_alpha_lookup -> trace_rays_batch -> render_lensed_image, bit for bit."""
    H, W = 120, 168
    vfov = np.radians(40.0)
    fov = (2 * np.arctan(np.tan(vfov / 2) * W / H), vfov)
    psi = (0.03, -0.04)
    bg = image_lens.synthetic_background(H, W, 7)
    metric = metrics.Schwarzschild(1.0) if a == 0 else metrics.Kerr(1.0, a, integrator="rk4", precision=32)
    al = image_lens.build_alpha_lookup((H, W), fov, psi=psi)
    if a == 0:
        fa, wd, _, _ = image_lens.precompute_final_alpha_lookup(al, 0.0, 100.0, metric)
    else:
        fa, wd, _, _ = image_lens.precompute_final_alpha_lookup_2d(al, fov, 0.0, 100.0, metric, psi=psi)
    staged = image_lens.render_lensed_image(bg, al, fa, wd, 0.0, fov, psi=psi)
    fused = image_lens.render_frame(bg, metric, 100.0, fov, psi=psi)
    assert np.array_equal(fused["fa"], fa, equal_nan=True)
    np.testing.assert_array_equal(fused["winding"], wd)
    np.testing.assert_array_equal(fused["rgb"], staged)


def test_drop_in_defaults_are_the_reference_arithmetic(tmp_path):
    """Switching to this package without naming a backend knob must give the reference's numbers: the plugin classes
    default to float64 (the reference has no float32 path, metrics.py:831-833) and, for Kerr, to its production
    integrator.  `image_lens.main(a=0)` with nothing but an image is checked against the reference's own lookup and
    picture of the same frame (F4 / F5) at the float64 tolerance."""
    import matplotlib.image as mpimg
    g, m = _g("48x64_a0")
    assert metrics.Schwarzschild().precision == 64 and metrics.Kerr(1.0, 0.9).precision == 64
    assert metrics.Kerr(1.0, 0.9).integrator == "dp45_exact"
    png = tmp_path / "bg.png"
    mpimg.imsave(str(png), g["background"])                       # 8-bit exact: the fixture's texture is uint8 / 255
    img = image_lens.main(a=0.0, r_obs_mult=m["r_obs"], image_path=str(png), output_path=str(tmp_path / "out.png"))
    fov = (m["hfov"], m["vfov"])
    out = image_lens.render_frame(g["background"], metrics.Schwarzschild(), m["r_obs"], fov, want=("fa", "winding", "rgb"))
    np.testing.assert_array_equal(img, out["rgb"])                # main() renders with the metric's defaults
    assert np.array_equal(np.isnan(out["fa"]), np.isnan(g["final_alpha"]))
    assert np.nanmax(np.abs(out["fa"] - g["final_alpha"])) <= 1e-6   # float64 kernel vs the reference's float64 tracer
    np.testing.assert_array_equal(out["winding"], g["winding"])
    assert (np.abs(img - g["lensed"]).max(axis=-1) > 0).mean() <= 0.005   # nearest-neighbour: a 1e-7 rad change can move a texel
    # the explicit fast path still exists and is float32
    fast = image_lens.render_frame(g["background"], metrics.Schwarzschild(precision=32), m["r_obs"], fov, want=("fa",))
    assert 1e-9 < np.nanmax(np.abs(fast["fa"] - g["final_alpha"])) <= 2e-4


def test_alpha_dedup_and_lookup_cache(tmp_path, capsys):
    """tracing every distinct alpha once gives byte-identical Schwarzschild lookups; a lookup cache written by
    one call is reused by the next call with the same metric, observer and camera (and by no other)."""
    g, m = _g("48x64_a0")
    S = metrics.Schwarzschild()
    full = image_lens.precompute_final_alpha_lookup(g["alpha_lookup"], m["alpha_crit"], m["r_obs"], S)
    dd = image_lens.precompute_final_alpha_lookup(g["alpha_lookup"], m["alpha_crit"], m["r_obs"], S, dedup=True)
    assert np.array_equal(full[0], dd[0], equal_nan=True) and np.array_equal(full[1], dd[1])
    assert full[2] == dd[2] == 48 * 64 and full[3] == 48 * 64 and dd[3] == np.unique(g["alpha_lookup"]).size < 48 * 64 / 3
    cache = str(tmp_path / "lookup_cache.npz")
    kw = dict(a=0.9, r_obs_mult=100.0, synthetic=(96, 64), lookup_cache=cache)
    first = image_lens.main(output_path=str(tmp_path / "a.png"), **kw)
    assert "cache hit" not in capsys.readouterr().out and os.path.exists(cache)
    second = image_lens.main(output_path=str(tmp_path / "b.png"), **kw)
    assert "Lookup cache hit" in capsys.readouterr().out
    np.testing.assert_array_equal(first, second)
    other = image_lens.main(output_path=str(tmp_path / "c.png"), **dict(kw, a=0.5))     # another metric: a miss, then rewritten
    bare = str(tmp_path / "cache_without_extension")            # the file is written under the name given, whatever it is
    image_lens.main(output_path=str(tmp_path / "f.png"), **dict(kw, lookup_cache=bare))
    capsys.readouterr()
    image_lens.main(output_path=str(tmp_path / "g.png"), **dict(kw, lookup_cache=bare))
    assert "Lookup cache hit" in capsys.readouterr().out and os.path.exists(bare)
    assert "cache hit" not in capsys.readouterr().out and not np.array_equal(other, first)
    staged = image_lens.main(output_path=str(tmp_path / "d.png"), staged=True, dedup_alpha=True, a=0.0, r_obs_mult=100.0, synthetic=(96, 64))
    plain = image_lens.main(output_path=str(tmp_path / "e.png"), staged=True, a=0.0, r_obs_mult=100.0, synthetic=(96, 64))
    np.testing.assert_array_equal(staged, plain)


def test_cli_main_runs_end_to_end(tmp_path, capsys):
    out = tmp_path / "lensed.png"
    img = image_lens.main(a=0.9, r_obs_mult=100.0, synthetic=(96, 64), output_path=str(out))
    text = capsys.readouterr().out
    assert img.shape == (64, 96, 3) and out.exists()
    assert "Metric: Kerr (M=1.0, a=0.9)" in text and "Benchmark summary" in text and "total rays: 6,144" in text
    assert "traced rays: 3,072" in text          # the reference's top-half trace + mirror is the default (image_lens.py:218-220)
    # the fused path writes the GPU's RGBA8 with its own PNG encoder: same pixels as the reference's imsave call
    import matplotlib.image as mpimg
    ref_png = tmp_path / "imsave.png"
    mpimg.imsave(str(ref_png), img)
    np.testing.assert_array_equal(mpimg.imread(str(out)), mpimg.imread(str(ref_png)))
    staged =image_lens.main(a=0.9, r_obs_mult=100.0, synthetic=(96, 64), output_path=str(out), staged=True)
    # staged mode applies the reference's top/bottom mirror, the fused default does not: compare the top half
    np.testing.assert_array_equal(staged[:32], img[:32])


def test_cli_default_is_the_reference_mirror_and_flags(tmp_path, capsys, monkeypatch):
    """`python image_lens.py --a 0.9` must give the reference's picture: the reference traces the top half and mirrors
    it (image_lens.py:218-220, :272-276, off by one row: quirk Q1), which only the staged path used to reproduce.
    Default fused == staged; --full-trace differs from it exactly where the mirror is off by one; --gpus 2 (rehearsed
    on one device) equals the full trace; an RGBA background is lensed by its colour planes."""
    kw = dict(a=0.9, r_obs_mult=100.0, synthetic=(96, 64), integrator="rk4", precision=32)
    default = image_lens.main(output_path=str(tmp_path / "a.png"), **kw)
    text = capsys.readouterr().out
    assert "bottom half mirrored as in the reference" in text
    staged = image_lens.main(output_path=str(tmp_path / "b.png"), staged=True, **kw)
    np.testing.assert_array_equal(default, staged)
    full = image_lens.main(output_path=str(tmp_path / "c.png"), full_trace=True, **kw)
    assert "every row traced" in capsys.readouterr().out
    assert np.array_equal(full[:32], default[:32]) and not np.array_equal(full[33:], default[33:])
    monkeypatch.setenv("LT_MULTI_DEVICES", "0,0")
    two = image_lens.main(output_path=str(tmp_path / "d.png"), gpus=2, **kw)
    assert "on 2 GPU(s)" in capsys.readouterr().out
    np.testing.assert_array_equal(two, full)
    # RGBA background
    import matplotlib.image as mpimg
    bg = image_lens.synthetic_background(64, 96, 0)
    rgba = np.concatenate([bg, np.ones((64, 96, 1), np.float32)], axis=-1)
    mpimg.imsave(str(tmp_path / "bg.png"), rgba)
    from_png = image_lens.main(a=0.9, r_obs_mult=100.0, image_path=str(tmp_path / "bg.png"), output_path=str(tmp_path / "e.png"),
                               integrator="rk4", precision=32)
    assert "alpha channel" in capsys.readouterr().out and from_png.shape == (64, 96, 3)
    planes = np.ascontiguousarray(mpimg.imread(str(tmp_path / "bg.png"))[..., :3])     # what the PNG round trip kept
    vfov = np.radians(40.0)
    fov = (2 * np.arctan(np.tan(vfov / 2) * 96 / 64), vfov)
    same = image_lens.render_frame(planes, metrics.Kerr(1.0, 0.9, integrator="rk4", precision=32), 100.0, fov, tb_symmetry=True)
    np.testing.assert_array_equal(from_png, same["rgb"])
    rgba_in = image_lens.render_frame(np.concatenate([planes, np.ones((64, 96, 1), np.float32)], -1),
                                      metrics.Kerr(1.0, 0.9, integrator="rk4", precision=32), 100.0, fov, tb_symmetry=True)
    np.testing.assert_array_equal(rgba_in["rgb"], same["rgb"])


def test_traced_shadow_matches_oracle():
    S = metrics.Schwarzschild(1.0)
    img, status, stats = black_hole_shadow.render_traced(S, 256, 256)
    al, _, _ = oracle.pixel_angles(256, 256, np.radians(40.0), np.radians(40.0))
    assert ((img == 0) != (al.astype(np.float64) < S.alpha_crit(50.0))).sum() <= 4
    assert stats["rays"] == 256 * 256
    K = metrics.Kerr(1.0, 0.9, integrator="rk4", precision=32)
    imgk, statusk, _ = black_hole_shadow.render_traced(K, 128, 128)
    ref = oracle.lookup("kerr", 1.0, 0.9, 50.0, 128, 128, np.radians(40.0), np.radians(40.0), integrator="rk4")
    assert ((statusk == 1) != (ref["status"] == 1)).sum() <= 16


@pytest.mark.parametrize("name", ["4x6_a0p9", "48x64_a0p9", "48x64_a0p9_psi", "33x40_a0p9"])
def test_production_pipeline_matches_reference_lookups(name):
    """The reference's default Kerr pipeline end to end -- DP45 float64, TB mirror, axis-refine
    columns, psi offsets -- against the lookups and images the reference itself produced (F4/F5)."""
    g, m = _g(name)
    fov, psi = (m["hfov"], m["vfov"]), tuple(m["psi"])
    K = metrics.Kerr(1.0, m["a"], integrator="dp45")
    al = image_lens.build_alpha_lookup((m["h"], m["w"]), fov, psi=psi)
    fa, wd, total, traced = image_lens.precompute_final_alpha_lookup_2d(al, fov, m["alpha_crit"], m["r_obs"], K, psi=psi)
    assert (total, traced) == (m["total"], m["traced"])
    nan_same = np.isnan(fa) == np.isnan(g["final_alpha"])
    assert (~nan_same).sum() <= 1
    both = ~np.isnan(fa) & ~np.isnan(g["final_alpha"])
    assert np.max(np.abs(fa[both] - g["final_alpha"][both])) <= 5e-6
    assert (wd[nan_same] != g["winding"][nan_same]).sum() <= 1
    img = image_lens.render_lensed_image(g["background"], al, fa, wd, m["alpha_crit"], fov, psi=psi)
    assert np.all(img == g["lensed"], axis=-1).mean() >= 0.995
    # fused call, same settings
    fused = image_lens.render_frame(g["background"], K, m["r_obs"], fov, psi=psi, tb_symmetry=True)
    assert np.array_equal(fused["fa"], fa, equal_nan=True) and np.array_equal(fused["rgb"], img)
