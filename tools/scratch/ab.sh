#!/bin/bash
# A/B of two library builds on one box, interleaved: tools/scratch/ab.sh libA.so libB.so   (files under .variants/)
for rep in 1 2 3; do for l in "$@"; do
  for sz in 2048 4096; do LTRACE_LIB=$PWD/.variants/$l python bench.py --size $sz --no-cpu-baseline --no-extras --steps 20 | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); print('$l size', d['config']['rays_per_frame'], d['value'], d['roofline']['avg_launch_ms'], d['config']['escaped'], d['config']['captured'])"; done
done; done
