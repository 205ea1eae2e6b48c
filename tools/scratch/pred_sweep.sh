#!/bin/bash
# k_dense_predict's time against its two hand-out parameters (LT_DENSE_PRED_REFILL, LT_DENSE_PRED_CHUNK), kernel trace per setting
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/pred_sweep; rm -rf "$OUT"; mkdir -p "$OUT"
for cfg in "8 64" "4 64" "2 64" "16 64" "24 64" "8 128" "8 32" "1 64"; do
  set -- $cfg
  export LT_DENSE_PRED_REFILL=$1 LT_DENSE_PRED_CHUNK=$2
  d="$OUT/r$1_c$2"
  rocprofv3 --kernel-trace --stats --output-format csv -d "$d" -- python3 tools/dense_bench.py 4194304 224 0.9 1 > "$d.json" 2> "$d.log" || exit 1
  python3 - "$d" "$cfg" <<'PY'
import csv, glob, sys
d, cfg = sys.argv[1], sys.argv[2]
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    rows = {}
    for r in csv.DictReader(open(f)):
        rows.setdefault(r["Kernel_Name"].split("(")[0], []).append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
    out = []
    for k, v in rows.items():
        v = sorted(v)[1:]
        if v: out.append(f"{k.split('::')[-1][:24]} {sum(e - s for s, e in v) / len(v) / 1e3:9.1f} us")
    print(f"refill {cfg.split()[0]:>2} chunk {cfg.split()[1]:>3}: " + "  ".join(sorted(out)))
PY
done | tee "$OUT/summary.txt"
