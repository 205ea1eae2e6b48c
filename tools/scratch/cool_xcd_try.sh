#!/bin/bash
# LT_COOL_XCD=1: chain-bound launches take k_kerr_direct<COOL> (long rays move to servers on XCD 0, lt_kernels.hpp).  Off / on, interleaved.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for mode in 0 1 0 1; do
  export LT_COOL_XCD=$mode
  echo "== LT_COOL_XCD=$mode"
  timeout -k 10 120 python3 bench.py --size 2048 --steps 20 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | python3 -c "
import sys, json; d = json.loads(sys.stdin.readline()); print('  2048^2        ', d['value'], 'Mrays/s', d['ms_per_step'], 'ms; K2', d['roofline']['avg_launch_ms'])" || exit 1
  for n in 8 4 2; do
    timeout -k 10 120 python3 bench.py --emulate-parts $n --emulate-part 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | python3 -c "
import sys, json; d = json.loads(sys.stdin.readline()); print('  rank 1 of $n     ', d['ms_per_step'], 'ms; K2', d['roofline']['avg_launch_ms'])" || exit 1
  done
  timeout -k 10 120 python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras 2>/dev/null | python3 -c "
import sys, json; d = json.loads(sys.stdin.readline()); print('  4096^2        ', d['value'], 'Mrays/s', d['ms_per_step'], 'ms; K2', d['roofline']['avg_launch_ms'])" || exit 1
done
