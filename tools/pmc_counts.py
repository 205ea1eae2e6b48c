#!/usr/bin/env python3
"""Measure, on the GPU box, what bench.py's roofline needs from the PMC counters and cannot read in-process:

  * SQ_INSTS_VALU of the integrate kernel per launch (executed wave-instructions) -> profiles/valu_counts.json
  * FETCH_SIZE x 2 + WRITE_SIZE of the integrate kernel per launch (HBM bytes; gfx950 correction of
    MI355X_MICROARCH.md: FETCH_SIZE reports half of a wide coalesced read) -> profiles/hbm_traffic.json

Each figure is keyed by bench.py's `config.profile_key` and stamped with the library's build id; bench.py refuses
a figure whose build id is not the loaded library's.  One rocprofv3 run per counter group (SQ, FETCH_SIZE,
WRITE_SIZE: separate passes, no trace domains besides --kernel-trace).

usage (through gpurun):  python3 tools/pmc_counts.py [--quick | --only=tag,tag]   (--quick: the two workloads of the driver's bench line)
                         -> gpurun_out/pmc_counts/{valu_counts,hbm_traffic}.json + summary.txt
then here:               cp gpurun_out/pmc_counts/*.json profiles/  (tools/refresh_profiles.sh collect does it)
"""
import csv
import glob
import json
import os
import subprocess
import sys

ROOT = os.environ.get("GRAFT_REPO_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "gpurun_out", "pmc_counts")
BENCH = os.path.join(ROOT, "bench.py")
COMMON = ["--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-extras"]

WORKLOADS = [
    ("north_star", []),
    ("north_star_queue", ["--schedule", "queue"]),
    ("kerr_2048", ["--size", "2048"]),
    ("image_lens_r100_bg", ["--r-obs", "100", "--background"]),
    ("dp45_exact_f64", ["--integrator", "dp45_exact", "--precision", "64"]),   # the production-path region of the bench line
    ("dp45_f64", ["--integrator", "dp45", "--precision", "64"]),
    ("kerr_8192_a099", ["--size", "8192", "--a", "0.99", "--steps", "2"]),
]
SQ = ["SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_ACTIVE_INST_VALU", "SQ_THREAD_CYCLES_VALU", "SQ_WAVES", "SQ_BUSY_CYCLES", "GRBM_GUI_ACTIVE"]


def profiled(tag, pmc, args):
    d = os.path.join(OUT, tag)
    subprocess.run(["rm", "-rf", d])
    os.makedirs(d, exist_ok=True)
    cmd = ["rocprofv3", "--kernel-trace", "--pmc"] + pmc + ["--output-format", "csv", "-d", d, "--",
                                                           "python3", BENCH] + COMMON + args
    env = dict(os.environ, TMPDIR="/tmp")
    r = subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True)
    line = [l for l in r.stdout.splitlines() if l.startswith("{")]
    if r.returncode != 0 or not line:
        print(f"!! {tag}: rc {r.returncode}\n{r.stderr[-2000:]}", flush=True)
        return None, {}
    sums, calls = {}, {}
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                k = row["Kernel_Name"]
                if "k_kerr_" not in k and "k_schw_rk4" not in k and "k_epilogue_frame" not in k and "k_prologue_camera" not in k:
                    continue
                kk = ("integrate" if ("k_kerr_" in k or "k_schw_rk4" in k) else ("epilogue" if "epilogue" in k else "prologue"), row["Counter_Name"])
                sums[kk] = sums.get(kk, 0.0) + float(row["Counter_Value"])
                calls[kk] = calls.get(kk, 0) + 1
    return json.loads(line[-1]), {k: v / calls[k] for k, v in sums.items()}


def main():
    os.makedirs(OUT, exist_ok=True)
    quick = "--quick" in sys.argv
    valu = {"workloads": {}}
    hbm = {"workloads": {}}
    text = []
    only = [a.split("=", 1)[1].split(",") for a in sys.argv if a.startswith("--only=")]
    todo = [w for w in WORKLOADS if w[0] in only[0]] if only else (
        [w for w in WORKLOADS if w[0] in ("north_star", "dp45_exact_f64")] if quick else WORKLOADS)
    # a partial run (--quick / --only=...) refreshes its workloads and keeps the other records (each carries its own
    # build id; bench.py marks a record measured on another build as stale and prints no traffic from it)
    for name, obj in (("valu_counts.json", valu), ("hbm_traffic.json", hbm)):
        try:
            with open(os.path.join(ROOT, "profiles", name)) as f:
                old = json.load(f)
            for k, rec in (old.get("workloads") or {}).items():
                rec.setdefault("build_id", old.get("build_id"))
                obj["workloads"][k] = rec
        except (OSError, ValueError):
            pass
    for tag, args in todo:
        j, c = profiled(tag + "_sq", SQ, args)
        if not j:
            continue
        key, bid = j["config"]["profile_key"], j["config"]["build_id"]
        valu["build_id"] = hbm["build_id"] = bid
        iters = j["roofline"]["executed"]["wave_iters_per_launch"]
        rec = {"build_id": bid, "valu_insts": int(c.get(("integrate", "SQ_INSTS_VALU"), 0)), "wave_iters": int(iters),
               "salu_insts": int(c.get(("integrate", "SQ_INSTS_SALU"), 0)),
               "lane_utilisation": round(c.get(("integrate", "SQ_THREAD_CYCLES_VALU"), 0) / max(c.get(("integrate", "SQ_ACTIVE_INST_VALU"), 1), 1) / 64, 4),
               "grbm_gui_active": int(c.get(("integrate", "GRBM_GUI_ACTIVE"), 0)),
               "launch_ms_under_profiler": j["roofline"]["avg_launch_ms"],
               "source": f"rocprofv3 --kernel-trace --pmc {' '.join(SQ)} -- python3 bench.py {' '.join(COMMON + args)} (tools/pmc_counts.py)"}
        valu["workloads"][key] = rec
        text.append(f"{key}\n   integrate kernel per launch: SQ_INSTS_VALU {rec['valu_insts']}  SQ_INSTS_SALU {rec['salu_insts']}  wave_iters {rec['wave_iters']}"
                    f"  -> {rec['valu_insts'] / max(rec['wave_iters'], 1):.2f} VALU per wave iteration; lane utilisation {rec['lane_utilisation']}; "
                    f"{rec['launch_ms_under_profiler']} ms under the profiler; clock (GRBM_GUI_ACTIVE/8/t) "
                    f"{rec['grbm_gui_active'] / 8 / max(rec['launch_ms_under_profiler'], 1e-9) / 1e3:.0f} MHz")
        jf, cf = profiled(tag + "_fetch", ["FETCH_SIZE"], args)
        jw, cw = profiled(tag + "_write", ["WRITE_SIZE"], args)
        if jf and jw and ("integrate", "FETCH_SIZE") in cf and ("integrate", "WRITE_SIZE") in cw:
            fetch_kb, write_kb = cf[("integrate", "FETCH_SIZE")], cw[("integrate", "WRITE_SIZE")]
            total = int(fetch_kb * 1024 * 2 + write_kb * 1024)
            hbm["workloads"][key] = {"build_id": bid, "bytes_per_launch": total, "fetch_size_kb": fetch_kb, "write_size_kb": write_kb,
                                     "algorithmic_bytes_per_launch": j["roofline"]["algorithmic_bytes_per_launch"],
                                     "source": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes), FETCH_SIZE x 2 "
                                               "(gfx950: it counts half of a wide coalesced read) + WRITE_SIZE, unit KB; tools/pmc_counts.py"}
            text.append(f"   HBM: FETCH_SIZE {fetch_kb:.1f} KB x2 + WRITE_SIZE {write_kb:.1f} KB = {total / 1e6:.1f} MB per launch; algorithmic "
                        f"{j['roofline']['algorithmic_bytes_per_launch'] / 1e6:.1f} MB; epilogue kernel FETCH x2 {cf.get(('epilogue', 'FETCH_SIZE'), 0) * 2048 / 1e6:.1f} MB, "
                        f"WRITE {cw.get(('epilogue', 'WRITE_SIZE'), 0) * 1024 / 1e6:.1f} MB")
        if "background" in " ".join(args):
            # the epilogue's background gather: L2 hit rate and fetched bytes, both sampling paths
            for mode in ("global", "lds"):
                jt, ct = profiled(tag + "_tcc_" + mode, ["TCC_HIT_sum", "TCC_MISS_sum"], args + ["--bg-sampling", mode])
                jf2, cf2 = profiled(tag + "_fetch_" + mode, ["FETCH_SIZE"], args + ["--bg-sampling", mode])
                if jt and jf2:
                    hit, miss = ct.get(("epilogue", "TCC_HIT_sum"), 0), ct.get(("epilogue", "TCC_MISS_sum"), 0)
                    text.append(f"   epilogue kernel, bg_sampling={mode}: {jt['roofline']['other_kernels_ms']['epilogue']} ms; L2 hit rate "
                                f"{hit / max(hit + miss, 1):.3f} (TCC_HIT_sum {hit:.0f}, TCC_MISS_sum {miss:.0f}); FETCH_SIZE x2 "
                                f"{cf2.get(('epilogue', 'FETCH_SIZE'), 0) * 2048 / 1e6:.1f} MB (records 536.9 MB + background 201.3 MB algorithmic); "
                                f"16x16 tiles staged in LDS / global fallback: {jt['config'].get('bg_groups_lds')} / {jt['config'].get('bg_groups_global')}")
        for name, obj in (("valu_counts.json", valu), ("hbm_traffic.json", hbm)):
            with open(os.path.join(OUT, name), "w") as f:
                json.dump(obj, f, indent=1)
        with open(os.path.join(OUT, "summary.txt"), "w") as f:
            f.write("\n".join(text) + "\n")
        print("\n".join(text[-4:]), flush=True)


if __name__ == "__main__":
    main()
