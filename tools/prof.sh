#!/bin/bash
# tools/prof.sh TAG [bench args...] -- bench.py under rocprofv3 on the GPU box, in a form whose numbers can be checked
# against the bench line's own HIP-event time:
#   plain A : the command WITHOUT the profiler                   -> avg_launch_ms as the driver sees it
#   trace   : rocprofv3 --kernel-trace --stats                   -> per-launch durations; the warm-up launches are
#                                                                   dropped from the statistics (tools/prof_summary.py)
#   plain B : without the profiler again, same lease             -> box drift between A and B bounds what is noise
#   pmc1/2  : --pmc ... (own runs, only --kernel-trace beside them)
# K = 20 timed launches after W = 5 warm-up launches (the driver's own command line; the chip needs ~5 launches to settle:
# 13.2, 11.9, 11.6, 11.3, 11.2, 11.17 ms in round 2's trace); only the standard launches (--no-extras), so every launch of the
# integrate kernel in the trace after the first W is one of the K the bench line averages.
# Summaries land in gpurun_out/prof_TAG/summary.txt ; tools/refresh_profiles.sh collect copies them into profiles/.
set -u
TAG=${1:-run}; shift || true
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG; rm -rf "$OUT"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
K=${PROF_STEPS:-20}; W=${PROF_WARMUP:-5}
ARGS="--steps $K --warmup $W --no-cpu-baseline --no-extras $*"
echo "$ARGS" > "$OUT/args.txt"
python3 "$ROOT/bench.py" $ARGS > "$OUT/bench_plain_a.json" 2> "$OUT/bench_plain_a.err"; echo "plain A rc=$?"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 "$ROOT/bench.py" $ARGS > "$OUT/bench_trace.log" 2>&1
echo "trace rc=$?"
python3 "$ROOT/bench.py" $ARGS > "$OUT/bench_plain_b.json" 2> "$OUT/bench_plain_b.err"; echo "plain B rc=$?"
if [ "${PROF_PMC:-1}" = 1 ]; then
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU GRBM_GUI_ACTIVE --output-format csv -d "$OUT/pmc1" -- python3 "$ROOT/bench.py" $ARGS > "$OUT/bench_pmc1.log" 2>&1
echo "pmc1 rc=$?"
rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_SALU SQ_INSTS_VALU_TRANS SQ_ACTIVE_INST_SCA SQ_WAVE_CYCLES --output-format csv -d "$OUT/pmc2" -- python3 "$ROOT/bench.py" $ARGS > "$OUT/bench_pmc2.log" 2>&1
echo "pmc2 rc=$?"
fi
python3 "$ROOT/tools/prof_summary.py" "$OUT" "$W" > "$OUT/summary.txt" 2>&1
cat "$OUT/summary.txt"
