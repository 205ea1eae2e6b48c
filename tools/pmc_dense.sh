#!/bin/bash
# Memory-pipe counters of the dense-trajectory kernel, caller order against the length-binned launch (separate --pmc passes).
# usage: pmc_dense.sh [n] [max_points] [tag]  -> gpurun_out/pmc_dense[_tag]/summary.txt   (LTRACE_LIB selects a build)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
N=${1:-4194304}; MP=${2:-224}
OUT=$PWD/gpurun_out/pmc_dense${3:+_$3}; rm -rf "$OUT"; mkdir -p "$OUT"
export TMPDIR=/tmp
i=0
for CNT in "TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum" "TA_DATA_STALLED_BY_TC_CYCLES_sum TA_FLAT_WRITE_WAVEFRONTS_sum" \
           "SQ_INST_CYCLES_VMEM_WR SQ_INSTS_VMEM_WR SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "WRITE_SIZE" "TCC_EA_WRREQ_sum TCC_EA_WRREQ_64B_sum" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_WRITE_REQ_sum" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS"; do
  i=$((i+1))
  for B in -1 1; do
    rocprofv3 --kernel-trace --pmc $CNT --output-format csv -d "$OUT/p${i}_b$B" -- python3 tools/dense_bench.py $N $MP 0.9 $B > "$OUT/p${i}_b$B.json" 2> "$OUT/p${i}_b$B.log" || echo "pass $i b $B failed"
  done
done
python3 - "$OUT" > "$OUT/summary.txt" <<'PY'
import csv, glob, os, sys
from collections import defaultdict
out = sys.argv[1]
res = defaultdict(dict)
for d in sorted(glob.glob(os.path.join(out, "p*_b*"))):
    if not os.path.isdir(d): continue
    b = d.rsplit("_b", 1)[1]
    agg, calls = defaultdict(float), defaultdict(int)
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if "k_dense_tracks" in row["Kernel_Name"]:
                agg[row["Counter_Name"]] += float(row["Counter_Value"]); calls[row["Counter_Name"]] += 1
    for c in agg: res[c][b] = agg[c] / calls[c]
print(f"{'counter (k_dense_tracks, per launch)':44s} {'caller order':>18s} {'length-binned':>18s}")
for c in sorted(res): print(f"{c:44s} {res[c].get('-1', float('nan')):18.1f} {res[c].get('1', float('nan')):18.1f}")
PY
cat "$OUT/summary.txt"
