#!/bin/bash
# tools/prof.sh TAG [bench args...] -- run bench.py under rocprofv3 on the GPU box:
#   pass 1: --kernel-trace --stats   (per-kernel durations)
#   pass 2/3: --pmc ...              (VALU issue / lane utilisation / wait counters), own runs
# Summaries land in gpurun_out/prof_TAG/ ; copy the ones worth judging into profiles/.
set -u
TAG=${1:-run}; shift || true
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG; rm -rf "$OUT"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 5 --warmup 2 --no-cpu-baseline --no-extras $*"   # only the standard launches: the averages must be the bench line's
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 "$ROOT/bench.py" $ARGS > "$OUT/bench_trace.log" 2>&1
echo "trace rc=$?"
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU GRBM_GUI_ACTIVE --output-format csv -d "$OUT/pmc1" -- python3 "$ROOT/bench.py" $ARGS > "$OUT/bench_pmc1.log" 2>&1
echo "pmc1 rc=$?"
rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_SALU SQ_INSTS_VALU_TRANS SQ_ACTIVE_INST_SCA SQ_WAVE_CYCLES --output-format csv -d "$OUT/pmc2" -- python3 "$ROOT/bench.py" $ARGS > "$OUT/bench_pmc2.log" 2>&1
echo "pmc2 rc=$?"
python3 "$ROOT/tools/prof_summary.py" "$OUT" > "$OUT/summary.txt" 2>&1
cat "$OUT/summary.txt"
