#!/bin/bash
# interleaved A/B of builds (file names under lib/) on the headline frame:  ab.sh libA.so libB.so [bench args]
L=$PWD/light-path-tracer_amd/lib
A=$1; B=$2; shift 2
for rep in 1 2 3; do
  for lib in $A $B; do
    LTRACE_LIB=$L/$lib python bench.py --no-extras --no-cpu-baseline "$@" 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib  %.1f Mrays/s  %.3f ms  integrate %.3f ms' % (d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms']))"
  done
done
