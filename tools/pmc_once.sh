#!/bin/bash
# tools/pmc_once.sh TAG "COUNTERS" [bench args] -- one rocprofv3 --pmc pass over bench.py
TAG=$1; CNT=$2; shift 2
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_$TAG
mkdir -p "$OUT"; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc $CNT --output-format csv -d "$OUT/pmc1" -- python3 "$ROOT/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --no-extras "$@" > "$OUT/bench.log" 2>&1
echo "rc=$?"; python3 "$ROOT/tools/prof_summary.py" "$OUT" | grep -A12 "k_kerr_\|k_schw"
