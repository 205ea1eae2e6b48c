"""CPU-side tests of the product package: host physics of metrics.py against the reference's
scalar known answers, the C-ABI library's symbol table, the row partition, loud failure
without a GPU.  No GPU compute here."""
import ctypes
import json
import os
import re

import numpy as np
import pytest

import ltrace
import metrics

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def scalars(golden_dir):
    with open(os.path.join(golden_dir, "scalars.json")) as f:
        return json.load(f)


def test_library_exports_every_declared_symbol():
    """Every function include/ltrace.h declares must be exported by libltrace_hip.so."""
    hdr = open(os.path.join(ROOT, "include", "ltrace.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(lt_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 15
    lib = ctypes.CDLL(ltrace.LIB_PATH)
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in ltrace.h but not exported"
    assert declared == set(ltrace.SIGNATURES), "ctypes binding and header disagree"
    assert ltrace.load().lt_version() == 200
    assert re.fullmatch(r"[0-9a-f]{12}", ltrace.build_id())


def test_struct_layouts_match_header():
    # sizes the C compiler gives for the structs in ltrace.h (x86-64 SysV)
    assert ctypes.sizeof(ltrace.Camera) == 8 + 6 * 8
    assert ctypes.sizeof(ltrace.Metric) == 8 + 2 * 8
    assert ctypes.sizeof(ltrace.Opts) == 8 * 4 + 3 * 8 + 8 + 8 + 8 + 8
    assert ctypes.sizeof(ltrace.Stats) == 16 * 8 + 3 * 8


def test_row_partition_is_a_partition():
    for H, rb, n in [(4096, 16, 8), (33, 16, 2), (100, 7, 3), (5, 16, 8), (64, 16, 1)]:
        seen = np.zeros(H, dtype=int)
        for p in range(n):
            rows = ltrace.global_rows(H, rb, n, p)
            assert len(rows) == ltrace.local_rows(H, rb, n, p)
            assert np.all(np.diff(rows) > 0)
            seen[rows] += 1
        assert np.all(seen == 1)
    assert ltrace.local_rows(64, 16, 4, 7) == -1      # part out of range


@pytest.mark.skipif(ltrace.device_count() > 0, reason="GPU present")
def test_no_gpu_means_loud_failure_not_fallback():
    fa = np.full(4, np.nan)
    w = np.zeros(4, dtype=np.int64)
    with pytest.raises(ltrace.LtraceError) as ei:
        metrics.Schwarzschild(1.0).trace_rays_batch(50.0, np.linspace(0.05, 0.3, 4), fa, w)
    assert ei.value.code == ltrace.ERR_NO_DEVICE
    assert np.all(np.isnan(fa))                      # nothing computed behind our back
    with pytest.raises(ltrace.LtraceError):
        metrics.Kerr(1.0, 0.9).trace_ray(50.0, 0.15, 0.7)


def test_schwarzschild_scalars(scalars):
    S = metrics.Schwarzschild(1.0)
    for r, v in scalars["schw_alpha_crit"].items():
        assert S.alpha_crit(float(r)) == pytest.approx(v, rel=0, abs=1e-15)
    assert S.capture_radius() == scalars["schw_capture_radius"]
    for deg, b in scalars["schw_b"].items():
        assert S.viewing_angle_to_impact_parameter(np.radians(float(deg)), 50.0) == pytest.approx(b, abs=1e-13)
    s0 = S.initial_conditions(50.0, 0.15)
    np.testing.assert_allclose(s0, scalars["schw_ic8"], rtol=1e-14, atol=0)
    np.testing.assert_allclose(S.geodesic_equations(0.0, s0), scalars["schw_rhs8"], rtol=1e-13, atol=1e-18)
    assert S.is_spherically_symmetric and S.R_S == 2.0 and S.R_PHOTON == 3.0


def test_kerr_scalars(scalars):
    for a_key, rec in scalars["kerr"].items():
        K = metrics.Kerr(1.0, float(a_key))
        assert K.r_plus == pytest.approx(rec["r_plus"], abs=1e-15)
        assert K.capture_radius() == pytest.approx(rec["capture_radius"], abs=1e-15)
        for r, v in rec["alpha_crit"].items():
            assert K.alpha_crit(float(r)) == pytest.approx(v, abs=1e-14)
        assert K.alpha_crit(50.0, 1.0) == pytest.approx(rec["alpha_crit_incl"], abs=1e-14)
        assert K.viewing_angle_to_impact_parameter(0.15, 50.0) == pytest.approx(rec["b"], abs=1e-12)
        np.testing.assert_allclose(K._unstable_photon_r(), rec["photon_r"], rtol=1e-14)
        if "crit" in rec:
            np.testing.assert_allclose(K._critical_impact_params(), rec["crit"], rtol=1e-12, atol=1e-12)
    assert not metrics.Kerr.is_spherically_symmetric
    with pytest.raises(ValueError):
        metrics.Kerr(1.0, 1.5)                        # reference metrics.py:849-850
    with pytest.raises(ValueError):
        metrics.Kerr(1.0, 0.0)._critical_impact_params()


def test_kerr_8d_flow_matches_reference(scalars):
    K = metrics.Kerr(1.0, 0.9)
    s0 = K.initial_conditions(50.0, 0.15, 0.7)
    np.testing.assert_allclose(s0, scalars["kerr_ic8"], rtol=1e-13, atol=1e-15)
    got = K.geodesic_equations(0.0, scalars["kerr_ic8"])
    np.testing.assert_allclose(got, scalars["kerr_rhs8"], rtol=1e-9, atol=1e-15)
    assert K.geodesic_equations(0.0, [0, K.r_plus, 1.0, 0, -1, 0, 0, 1]) == [0.0] * 8


def test_png_writer_decodes_to_the_pixels_imsave_writes(tmp_path):
    """write_png_rgba8 (fast path for the GPU's RGBA8) against the reference's save call, mpimg.imsave of the
    float image (image_lens.py:510): both files must decode to identical pixels."""
    import matplotlib
    matplotlib.use("Agg")
    import matplotlib.image as mpimg
    import image_lens
    rng = np.random.default_rng(4)
    rgb = rng.random((37, 53, 3), dtype=np.float32)
    rgb[0, 0] = (0.0, 1.0, 0.999999)
    rgba = np.empty((37, 53, 4), dtype=np.uint8)
    rgba[..., :3] = (rgb * 255).astype(np.uint8)      # matplotlib's own float -> uint8 rule
    rgba[..., 3] = 255
    a, b = str(tmp_path / "a.png"), str(tmp_path / "b.png")
    mpimg.imsave(a, rgb)
    image_lens.write_png_rgba8(b, rgba)
    ia, ib = mpimg.imread(a), mpimg.imread(b)
    assert ia.shape == ib.shape == (37, 53, 4) and np.array_equal(ia, ib)
    with pytest.raises(ValueError):
        image_lens.write_png_rgba8(b, rgba[..., :3])


def test_row_partition_properties_hold_for_any_frame():
    """Property test of lt_local_rows / lt_global_row: for any height, block size and partition count the parts
    tile range(height) exactly once, each part's rows ascend, and blocks go round-robin."""
    from hypothesis import given, settings, strategies as st

    @settings(max_examples=150, deadline=None)
    @given(st.integers(1, 5000), st.integers(1, 64), st.integers(1, 16))
    def check(height, row_block, n_parts):
        seen = np.zeros(height, dtype=np.int32)
        for p in range(n_parts):
            rows = ltrace.global_rows(height, row_block, n_parts, p)
            assert rows.size == ltrace.local_rows(height, row_block, n_parts, p)
            if rows.size:
                assert np.all(np.diff(rows) > 0) and rows[0] >= 0 and rows[-1] < height
                assert np.all((rows // row_block) % n_parts == p)
            seen[rows] += 1
        assert np.all(seen == 1)

    check()
    assert ltrace.local_rows(10, 16, 4, 4) == -1 and ltrace.local_rows(0, 16, 1, 0) == -1   # out-of-range part, empty frame


def test_ctypes_structs_have_the_sizes_the_c_compiler_gives(tmp_path):
    """Compile a probe against include/ltrace.h with gcc and compare sizeof / offsetof with the ctypes mirrors."""
    import subprocess
    src = tmp_path / "probe.c"
    src.write_text(
        '#include <stdio.h>\n#include <stddef.h>\n#include "ltrace.h"\n'
        'int main(void) { printf("%zu %zu %zu %zu %zu %zu %zu %zu\\n", sizeof(lt_camera), sizeof(lt_metric), sizeof(lt_opts),\n'
        '  sizeof(lt_stats), sizeof(lt_dense_opts), offsetof(lt_opts, stream), offsetof(lt_dense_opts, max_points),\n'
        '  offsetof(lt_dense_opts, stream)); return 0; }\n')
    exe = tmp_path / "probe"
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.check_call(["gcc", "-std=c11", "-I", os.path.join(root, "include"), str(src), "-o", str(exe)])
    got = [int(x) for x in subprocess.check_output([str(exe)]).split()]
    want = [ctypes.sizeof(ltrace.Camera), ctypes.sizeof(ltrace.Metric), ctypes.sizeof(ltrace.Opts), ctypes.sizeof(ltrace.Stats),
            ctypes.sizeof(ltrace.DenseOpts), ltrace.Opts.stream.offset, ltrace.DenseOpts.max_points.offset,
            ltrace.DenseOpts.stream.offset]
    assert got == want


def test_design_md_tables_are_generated_from_the_profiles():
    """DESIGN.md's numeric tables are written by tools/design_tables.py from profiles/r03_bench_*.json: a figure that
    was refreshed under profiles/ but not in the document (or edited by hand in the document) fails here."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "design_tables.py"), "--check"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
