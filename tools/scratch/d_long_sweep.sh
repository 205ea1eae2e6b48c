#!/bin/bash
# LT_D_LONG (steps after which a wavefront enters its ghost-lane phase: lone-wave step form, issue priority), plain kernel:
# integrate-kernel ms of the 4096^2 frame and of chain-bound launches, two passes interleaved
cd "${GRAFT_REPO_ROOT:-/root/repo}"
run() { timeout -k 10 120 python3 bench.py "$@" --steps 12 --warmup 4 --no-cpu-baseline --no-extras 2>/dev/null | python3 -c "
import sys, json; d = json.loads(sys.stdin.readline()); print('%.3f' % d['roofline']['avg_launch_ms'], end=' ')" || exit 1; }
for pass in 1 2; do
for L in 1024 128 192 256 384 512 768 2048; do
  export LT_D_LONG=$L
  echo -n "pass $pass LT_D_LONG $L:  K2 ms  4096^2 / 2048^2 / rank of 8 / rank of 4 / rank of 2:  "
  run; run --size 2048; run --emulate-parts 8 --emulate-part 1; run --emulate-parts 4 --emulate-part 1; run --emulate-parts 2 --emulate-part 1; echo
done
done
