#!/bin/bash
# LT_D_LONG (iterations after which a wave of the direct kernel raises its priority and keeps its finished lanes enabled as ghosts)
for d in 64 128 256 512 1024 2048; do
  for args in "--size 4096" "--size 2048" "--size 4096 --emulate-parts 8 --emulate-part 0"; do
    LT_D_LONG=$d python bench.py $args --no-extras --steps 30 --warmup 5 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('LT_D_LONG=$d  $args:  %.1f Mrays/s  %.3f ms' % (d['value'], d['ms_per_step']))"
  done
done
