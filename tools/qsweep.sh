#!/bin/bash
# sweep queue-kernel knobs (env) on the benchmark frame; prints integrate-kernel ms
run() { echo -n "$* : "; env "$@" timeout -k 10 200 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --schedule queue | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['roofline']['avg_launch_ms'], d['value'])"; }
run LT_Q_BPC=8
run LT_Q_BPC=6
run LT_Q_BPC=4
run LT_Q_BPC=3
run LT_Q_BPC=2
run LT_Q_REFILL=1
run LT_Q_REFILL=8
run LT_Q_REFILL=16
run LT_Q_REFILL=32
run LT_Q_CHUNK=256
run LT_Q_CHUNK=1024
run LT_Q_LONG=100000
run LT_Q_LONG=300
