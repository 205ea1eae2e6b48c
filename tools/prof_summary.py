#!/usr/bin/env python3
"""Condense a tools/prof.sh output directory: per-kernel duration stats from the kernel trace,
per-kernel PMC sums.  Prints a small text table (what gets committed under profiles/)."""
import csv
import glob
import os
import sys
from collections import defaultdict

out = sys.argv[1]


def find(sub, pat):
    return sorted(glob.glob(os.path.join(out, sub, "**", pat), recursive=True))


def short(name):
    name = name.replace("void ", "")
    return name.split("(")[0][:60]


for f in find("trace", "*kernel_stats.csv"):
    print("== kernel stats (rocprofv3 --kernel-trace --stats):", os.path.relpath(f, out))
    with open(f) as fh:
        for i, row in enumerate(csv.DictReader(fh)):
            if i == 0:
                print(f"{'kernel':60s} {'calls':>6s} {'avg_us':>12s} {'min_us':>12s} {'max_us':>12s} {'pct':>7s}")
            print(f"{short(row['Name']):60s} {row['Calls']:>6s} {float(row['AverageNs'])/1e3:12.2f} "
                  f"{float(row['MinNs'])/1e3:12.2f} {float(row['MaxNs'])/1e3:12.2f} {row['Percentage']:>7s}")

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
