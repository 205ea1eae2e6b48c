#!/bin/bash
# hand-off threshold (LT_D_LONG) and server count (LT_COOL_SERVERS) of k_kerr_direct<COOL>, chain-bound launches
cd "${GRAFT_REPO_ROOT:-/root/repo}"
run() { timeout -k 10 120 python3 bench.py "$@" --steps 12 --warmup 4 --no-cpu-baseline --no-extras 2>/dev/null | python3 -c "
import sys, json; d = json.loads(sys.stdin.readline()); print('%.3f' % d['roofline']['avg_launch_ms'], end=' ')" || exit 1; }
for cfg in "0 1024 16" "1 1024 16" "1 768 16" "1 512 16" "1 512 64" "1 384 64" "1 384 128" "1 256 128" "1 256 512" "0 512 16" "0 256 16"; do
  set -- $cfg
  export LT_COOL_XCD=$1 LT_D_LONG=$2 LT_COOL_SERVERS=$3
  echo -n "cool $1 long $2 servers $3:  K2 ms  2048^2 / rank of 8 / rank of 4:  "
  run --size 2048; run --emulate-parts 8 --emulate-part 1; run --emulate-parts 4 --emulate-part 1; echo
done
