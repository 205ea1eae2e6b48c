#!/bin/bash
# interleaved A/B of library builds on the dense benchmark:  dense_ab.sh "libA.so:binning libB.so:binning ..." [n] [max_points]
cd "${GRAFT_REPO_ROOT:-/root/repo}"
N=${2:-4194304}; MP=${3:-224}
for rep in 1 2 3; do
  for spec in $1; do
    lib=${spec%%:*}; b=${spec##*:}
    LTRACE_LIB=$PWD/$lib python3 tools/dense_bench.py $N $MP 0.9 $b 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib binning $b  %.3f ms  %.1f M tracks/s' % (d['ms'], d['tracks_per_s'] / 1e6))"
  done
done
