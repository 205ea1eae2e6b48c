#!/bin/bash
# Frames in flight: what overlapping the tail of one frame with the bulk of the next buys (bench.py --frames-in-flight F),
# for the whole frame on one GPU and for what ONE rank of an N-GPU run renders (--emulate-parts N).
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
run() {
    python3 bench.py --no-cpu-baseline --no-extras --steps 20 --warmup 4 "$@" 2>/dev/null | tail -1 | python3 -c "
import sys, json
d = json.loads(sys.stdin.readline()); r = d['roofline']; c = d['config']
print(f\"{' '.join(sys.argv[1:]):60s} {d['value']:9.1f} Mrays/s  {d['ms_per_step']:8.4f} ms/frame  integrate launch {r['avg_launch_ms']:8.4f} ms  frac {r['frac']}  region {r.get('region_issue_frac')}\")" "$@"
}
for F in 1 2 3; do run --size 4096 --frames-in-flight $F; done
for F in 1 2 3; do run --size 2048 --frames-in-flight $F; done
for N in 2 4 8; do for F in 1 2 3; do run --size 4096 --emulate-parts $N --emulate-part 1 --frames-in-flight $F; done; done
for F in 1 2; do run --size 4096 --integrator dp45 --precision 64 --frames-in-flight $F; done
