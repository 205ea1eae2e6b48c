#!/bin/bash
# dense-track kernel A/B: "lib[:ghost_at]" arguments (file names under lib/, LT_DENSE_GHOST_AT), interleaved
L=$PWD/light-path-tracer_amd/lib
for rep in 1 2; do
  for spec in "$@"; do
    lib=${spec%%:*}; g=${spec#*:}; [ "$g" = "$spec" ] && g=8
    LT_DENSE_GHOST_AT=$g LTRACE_LIB=$L/$lib python tools/dense_bench.py 1048576 256 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib ghost_at=$g  1M x 256: %.3f ms  %.1f M tracks/s' % (d['ms'], d['tracks_per_s'] / 1e6))"
  done
done
