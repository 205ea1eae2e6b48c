#!/usr/bin/env python3
"""Condense a tools/prof.sh output directory into the text kept under profiles/:

  * per-kernel durations from the kernel TRACE (every dispatch), with the first W launches of each kernel -- the
    warm-up frames -- left out of the statistics, and the per-launch microseconds of the integrate kernel listed;
  * the same run's bench line (HIP events under the profiler) and the two runs without the profiler on the same lease,
    so that profiler overhead and box drift are numbers, not guesses;
  * the roofline fraction recomputed from THIS file's average (SQ_INSTS_VALU of pmc1 / issue peak / trace average);
  * rocprofv3's own --stats table (which includes the warm-up launches) for reference;
  * per-kernel PMC sums of the pmc passes.
usage: prof_summary.py DIR [warmup_launches]"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

out = sys.argv[1]
warm = int(sys.argv[2]) if len(sys.argv) > 2 else 0
N_SIMD, CLOCK = 1024, 2.4e9


def find(sub, pat):
    return sorted(glob.glob(os.path.join(out, sub, "**", pat), recursive=True))


def short(name):
    name = name.replace("void ", "")
    return name.split("(")[0][:64]


def bench_line(path):
    try:
        with open(path) as f:
            lines = [l for l in f.read().splitlines() if l.startswith("{")]
        return json.loads(lines[-1]) if lines else None
    except OSError:
        return None


def is_ours(k):
    return k.startswith("lt::")


try:
    print("# bench.py", open(os.path.join(out, "args.txt")).read().strip())
except OSError:
    pass

# ---- kernel trace: every dispatch, warm-up launches dropped
per = defaultdict(list)
for f in find("trace", "*kernel_trace.csv"):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            per[short(row["Kernel_Name"])].append((int(row["Start_Timestamp"]), int(row["End_Timestamp"])))
integ_avg_us = None
if per:
    print(f"== kernel trace (rocprofv3 --kernel-trace): first {warm} launches of each kernel (warm-up frames) excluded")
    print(f"{'kernel':64s} {'launches':>8s} {'avg_us':>11s} {'min_us':>11s} {'max_us':>11s}")
    for k, v in sorted(per.items(), key=lambda kv: -sum(e - s for s, e in kv[1])):
        if not is_ours(k):
            continue
        v.sort()
        d = [(e - s) / 1e3 for s, e in v[warm:]] or [(e - s) / 1e3 for s, e in v]
        print(f"{k:64s} {len(d):8d} {sum(d) / len(d):11.2f} {min(d):11.2f} {max(d):11.2f}")
        if "k_kerr_" in k or "k_schw_rk4" in k or "k_dense_tracks" in k:
            if integ_avg_us is None:
                integ_avg_us = sum(d) / len(d)
                integ_name, integ_list = k, d
    if integ_avg_us is not None:
        print(f"   per launch, {integ_name}: " + " ".join(f"{x:.0f}" for x in integ_list) + "  us")

# ---- the bench lines: under the profiler, and without it before / after on the same lease
prof = bench_line(os.path.join(out, "bench_trace.log"))
pa, pb = bench_line(os.path.join(out, "bench_plain_a.json")), bench_line(os.path.join(out, "bench_plain_b.json"))
rows = [("without profiler, before", pa), ("under rocprofv3 --kernel-trace --stats", prof), ("without profiler, after", pb)]
if any(r for _, r in rows):
    print("== bench.py's own HIP-event average of the integrate kernel (roofline.avg_launch_ms), same command, same lease")
    for label, d in rows:
        if d:
            r = d["roofline"]
            ex = r.get("executed", {})
            print(f"   {label:42s} avg_launch_ms {r['avg_launch_ms']:.4f}  ms_per_step {d['ms_per_step']:.4f}  frac {r.get('frac')}"
                  f"  held clock {ex.get('clock_mhz_held')} MHz  frac_at_held_clock {r.get('frac_at_held_clock') or ex.get('frac_at_held_clock')}")
    if integ_avg_us and prof:
        ev = prof["roofline"]["avg_launch_ms"] * 1e3
        print(f"   trace average {integ_avg_us:.1f} us vs HIP events in the same run {ev:.1f} us: {100 * (integ_avg_us / ev - 1):+.2f} %")
    plain = [d["roofline"]["avg_launch_ms"] for d in (pa, pb) if d]
    if integ_avg_us and plain:
        m = sum(plain) / len(plain) * 1e3
        print(f"   trace average vs mean of the runs without the profiler ({m:.1f} us): {100 * (integ_avg_us / m - 1):+.2f} % "
              f"(profiler overhead + box drift; the two plain runs differ by {100 * abs(plain[0] / plain[-1] - 1):.2f} %)")

for f in find("trace", "*kernel_stats.csv"):
    print("== rocprofv3 --stats table as written (all launches, warm-up included):", os.path.relpath(f, out))
    with open(f) as fh:
        for i, row in enumerate(csv.DictReader(fh)):
            if i == 0:
                print(f"{'kernel':64s} {'calls':>6s} {'avg_us':>12s} {'min_us':>12s} {'max_us':>12s} {'pct':>7s}")
            if is_ours(short(row["Name"])):
                print(f"{short(row['Name']):64s} {row['Calls']:>6s} {float(row['AverageNs'])/1e3:12.2f} "
                      f"{float(row['MinNs'])/1e3:12.2f} {float(row['MaxNs'])/1e3:12.2f} {row['Percentage']:>7s}")

valu_per_launch = None
for sub in ("pmc1", "pmc2"):
    files = find(sub, "*counter_collection.csv")
    if not files:
        continue
    agg = defaultdict(lambda: defaultdict(float))
    calls = defaultdict(int)
    for f in files:
        with open(f) as fh:
            for row in csv.DictReader(fh):
                k = short(row["Kernel_Name"])
                if not is_ours(k):
                    continue
                agg[k][row["Counter_Name"]] += float(row["Counter_Value"])
                calls[(k, row["Counter_Name"])] += 1
    print(f"== PMC per launch ({sub}), averaged over launches")
    for k, cs in agg.items():
        print(" ", k)
        for c, v in sorted(cs.items()):
            n = calls[(k, c)]
            print(f"     {c:28s} {v / n:18.1f}  (x{n})")
        d = {c: v / calls[(k, c)] for c, v in cs.items()}
        if "SQ_ACTIVE_INST_VALU" in d and "SQ_THREAD_CYCLES_VALU" in d and d["SQ_ACTIVE_INST_VALU"]:
            print(f"     -> lane utilisation (THREAD_CYCLES_VALU / ACTIVE_INST_VALU / 64) = "
                  f"{d['SQ_THREAD_CYCLES_VALU'] / d['SQ_ACTIVE_INST_VALU'] / 64:.3f}")
        if "SQ_ACTIVE_INST_VALU" in d and "SQ_BUSY_CYCLES" in d and d["SQ_BUSY_CYCLES"]:
            print(f"     -> ACTIVE_INST_VALU / BUSY_CYCLES = {d['SQ_ACTIVE_INST_VALU'] / d['SQ_BUSY_CYCLES']:.3f}")
        if "SQ_INSTS_VALU" in d and ("k_kerr_" in k or "k_dense_tracks" in k) and valu_per_launch is None:
            valu_per_launch = d["SQ_INSTS_VALU"]

if valu_per_launch and integ_avg_us and prof:
    cyc = 2.0 if prof.get("dtype") == "f32" else 4.0
    peak = N_SIMD * CLOCK / cyc
    print("== roofline recomputed from THIS file")
    print(f"   SQ_INSTS_VALU {valu_per_launch:.0f} per launch (pmc1) / ({N_SIMD} SIMDs x 2.4 GHz / {cyc:g} cycles = {peak:.4e} wave-instructions/s)")
    print(f"   / trace average {integ_avg_us:.1f} us  ->  frac {valu_per_launch / (integ_avg_us * 1e-6) / peak:.4f}"
          f"   (bench line of the profiled run: {prof['roofline'].get('frac')}"
          + (f"; without the profiler: {', '.join(str(d['roofline'].get('frac')) for d in (pa, pb) if d)})" if (pa or pb) else ")"))
