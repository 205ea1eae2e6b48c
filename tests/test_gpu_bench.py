"""bench.py on the GPU box, as the driver runs it (a subprocess: torch is imported before the library there):
the JSON contract, the executed-work roofline (frac <= 1 by construction), the extras."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(args, timeout=600):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, env=env,
                       timeout=timeout)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


def test_north_star_line_has_the_contract_fields_and_an_honest_roofline():
    d = _bench(["--steps", "3", "--warmup", "1", "--cpu-seconds", "2"])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["metric"] == d["unit"] == "Mrays/s" and d["n_gpus"] == 1 and d["steps"] == 3 and d["dtype"] == "f32"
    assert d["config"]["workload"] == "kerr_a0.9_shadow_4096x4096_r50_rk4" and d["config"]["rays_per_frame"] == 4096 * 4096
    assert d["vs_baseline"] is None and d["scaling"] == "strong" and d["higher_is_better"] is True
    assert abs(d["value"] - d["config"]["rays_per_frame"] / d["ms_per_step"] / 1e3) < 0.01 * d["value"]
    r = d["roofline"]
    assert r["peak"] == 157.3 and r["unit"] == "TFLOP/s" and r["bound"] == "valu_issue_fp32"
    assert r["executed"]["wave_iters_per_launch"] > 4e7 and 1500 < r["executed"].get("clock_mhz_held", 2000) < 2600
    if r["frac"] is not None:                       # a VALU count for this workload is committed under profiles/
        assert 0.5 < r["frac"] <= 1.0 and abs(r["achieved"] / r["peak"] - r["frac"]) < 2e-3
        assert r["frac"] <= r["executed"]["frac_at_held_clock"] <= 1.0
        assert r["avg_launch_ms"] + r["other_kernels_ms"]["prologue"] + r["other_kernels_ms"]["epilogue"] <= d["ms_per_step"] * 1.02
    if r["traffic"] is not None:                    # only printed when the profile's build id is the library's
        assert 0.95 < r["traffic"] / r["algorithmic_bytes_per_launch"] < 1.2
    assert r["algorithmic_as_written"]["flops_per_launch"] > 2e12
    assert d["pipelined"]["frames_in_flight"] == 3 and d["pipelined"]["value"] > 0.9 * d["value"]
    assert d["chain_floor"]["longest_ray_steps"] > 2000 and d["chain_floor"]["alone_ms"] < r["avg_launch_ms"]
    assert 5.0 < d["end_to_end_ms"]["pinned_dst_ms"] < 40.0
    assert d["cpu_baseline"]["unit"] == "Mrays/s" and d["cpu_baseline"]["cores"] >= 1 and d["cpu_baseline"]["kind"] == "port" and "perf build" in d["cpu_baseline"]["build"]
    assert d["ranks"]["rccl_world"] == 1 and d["ranks"]["devices"][0]["rank"] == 0
    assert d["config"]["balance"] == {"requested": "auto", "used": "cyclic"}
    # the reference's production path (DP45 float64, the plugin default) is measured in the same line, with its own
    # roofline from a committed PMC count and its own CPU baseline
    p = d["production_path"]
    assert p["dtype"] == "f64" and p["steps"] == 3 and p["config"]["workload"] == "kerr_a0.9_shadow_4096x4096_r50_dp45_exact"
    assert p["config"]["rays_per_frame"] == 4096 * 4096 and 40 < p["config"]["mean_dp45_attempts_per_ray"] < 80
    assert abs(p["value"] - 4096 * 4096 / p["ms_per_step"] / 1e3) < 0.01 * p["value"]
    pr = p["roofline"]
    assert pr["bound"] == "valu_issue_fp64" and pr["peak"] == 78.6 and "Dp45" not in pr["kernel"] and "dp45_exact" in pr["kernel"]
    assert pr["frac"] is not None, "profiles/valu_counts.json has no record for the production-path workload"
    assert 0.5 < pr["frac"] <= pr["frac_at_held_clock"] <= 1.0
    assert pr["avg_launch_ms"] + pr["other_kernels_ms"]["prologue"] + pr["other_kernels_ms"]["epilogue"] <= p["ms_per_step"] * 1.02
    assert p["cpu_baseline"]["unit"] == "Mrays/s" and "dp45" in p["cpu_baseline"]["sample"]
    assert p["cpu_baseline"]["mean_rhs_evals_per_ray"] < d["cpu_baseline"]["mean_rhs_evals_per_ray"]     # adaptive: fewer evaluations
    # what each rank of a 2 / 4 / 8 GPU run renders, one at a time on this GPU (a projection, labelled as one)
    for integ in ("rk4", "dp45_exact"):
        pj = d["projected_ranks"][integ]
        assert "PROJECTION" in pj["what"]
        for n in (2, 4, 8):
            assert len(pj[str(n)]["frame_ms_per_rank"]) == n and pj[str(n)]["slowest_rank_ms"] == max(pj[str(n)]["frame_ms_per_rank"])
        assert pj["8"]["slowest_rank_ms"] < pj["2"]["slowest_rank_ms"] < d["ms_per_step"] * (1.0 if integ == "rk4" else 2.0)


def test_other_workloads_and_flags():
    d = _bench(["--size", "1024", "--metric", "schwarzschild", "--steps", "3", "--warmup", "1", "--no-cpu-baseline"])
    assert d["config"]["workload"].startswith("schwarzschild_a0.0_shadow_1024x1024") and d["roofline"]["frac"] is None
    d = _bench(["--size", "1024", "--r-obs", "100", "--background", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-extras"])
    assert d["config"]["workload"].endswith("_lensed_background") and d["config"]["bg_sampling"] == "global" and "pipelined" not in d
    d = _bench(["--size", "1024", "--frames-in-flight", "2", "--steps", "4", "--warmup", "2", "--no-cpu-baseline"])
    assert d["config"]["frames_in_flight"] == 2
    d = _bench(["--size", "1024", "--emulate-parts", "4", "--emulate-part", "3", "--steps", "2", "--warmup", "1",
                "--no-cpu-baseline", "--no-extras"])
    assert d["config"]["rays_per_frame"] == 1024 * 256 and "emulated rank 3" in d["config"]["row_partition"]
