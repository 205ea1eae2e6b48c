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
cat "$OUT/bench.json" >> "$OUT/summary.txt"
tail -40 "$OUT/summary.txt"
