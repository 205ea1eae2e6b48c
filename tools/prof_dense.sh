#!/bin/bash
# rocprofv3 kernel trace + one PMC pass of the dense-trajectory benchmark; summary -> gpurun_out/prof_dense[_TAG]/summary.txt
# usage: prof_dense.sh [TAG] [dense_bench.py args: n max_points a binning]
set -e
cd "${GRAFT_REPO_ROOT:-/root/repo}"
TAG=${1:-}; shift || true
OUT=$PWD/gpurun_out/prof_dense${TAG:+_$TAG}
rm -rf "$OUT"; mkdir -p "$OUT"
export TMPDIR=/tmp
echo "tools/dense_bench.py $*" > "$OUT/args.txt"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 tools/dense_bench.py "$@" > "$OUT/bench.json" 2> "$OUT/trace.log"
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d "$OUT/pmc1" -- python3 tools/dense_bench.py "$@" > "$OUT/bench_pmc1.json" 2> "$OUT/pmc1.log"
python3 tools/prof_summary.py "$OUT" 1 > "$OUT/summary.txt" 2>&1 || true     # 1 warm-up launch in dense_bench.py
# the executed-instruction record dense_bench.py prices k_dense_tracks with (profiles/valu_counts.json, keyed by workload, stamped with the build id)
python3 - "$OUT" "$@" <<'PY' || true
import csv, glob, json, os, sys
out, args = sys.argv[1], sys.argv[2:]
n, mp, a, b = (args + ["1048576", "256", "0.9", "0"])[:4] if len(args) < 4 else args[:4]
agg, calls = {}, {}
for f in glob.glob(os.path.join(out, "pmc1", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_dense_tracks" in r["Kernel_Name"]:
            agg[r["Counter_Name"]] = agg.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"]); calls[r["Counter_Name"]] = calls.get(r["Counter_Name"], 0) + 1
dur = []
for f in glob.glob(os.path.join(out, "trace", "**", "*kernel_trace.csv"), recursive=True):
    rows = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in csv.DictReader(open(f)) if "k_dense_tracks" in r["Kernel_Name"])
    dur += [(e - s) / 1e6 for s, e in rows[1:]]
if "SQ_INSTS_VALU" in agg and dur:
    sys.path.insert(0, os.path.join(os.getcwd(), "light-path-tracer_amd"))
    import ltrace
    c = {k: v / calls[k] for k, v in agg.items()}
    rec = {"build_id": ltrace.build_id(), "valu_insts": int(c["SQ_INSTS_VALU"]), "kernel_ms_in_the_profiled_run": round(sum(dur) / len(dur), 4),
           "lane_utilisation": round(c["SQ_THREAD_CYCLES_VALU"] / c["SQ_ACTIVE_INST_VALU"] / 64, 4),
           "source": f"rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU ... -- python3 tools/dense_bench.py {n} {mp} {a} {b} (tools/prof_dense.sh)"}
    json.dump({f"dense_tracks|kerr_a{float(a)}|n{int(n)}|mp{int(mp)}|binning{int(b)}": rec}, open(os.path.join(out, "valu_record.json"), "w"), indent=1)
    print("valu record:", rec)
PY
# the benchmark line once more WITHOUT the profiler, priced with the record just written (same build, same box)
LT_VALU_RECORD="$OUT/valu_record.json" python3 tools/dense_bench.py "$@" > "$OUT/bench_plain.json" 2>/dev/null || cp "$OUT/bench.json" "$OUT/bench_plain.json"
cat "$OUT/bench_plain.json" >> "$OUT/summary.txt"
tail -40 "$OUT/summary.txt"
