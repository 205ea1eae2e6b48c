#!/bin/bash
# rocprofv3 kernel trace + one PMC pass of the dense-trajectory benchmark; summary -> gpurun_out/prof_dense/summary.txt
set -e
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=$PWD/gpurun_out/prof_dense
rm -rf "$OUT"; mkdir -p "$OUT"
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 tools/dense_bench.py > "$OUT/bench.json" 2> "$OUT/trace.log"
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d "$OUT/pmc1" -- python3 tools/dense_bench.py > "$OUT/bench_pmc1.json" 2> "$OUT/pmc1.log"
python3 tools/prof_summary.py "$OUT" > "$OUT/summary.txt" 2>&1 || true
cat "$OUT/bench.json" >> "$OUT/summary.txt"
tail -30 "$OUT/summary.txt"
