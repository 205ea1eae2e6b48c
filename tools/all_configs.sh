#!/bin/bash
# One bench.py line per BASELINE.json config shape that fits one GPU (plus two beyond them), condensed.
# usage (on the GPU box): bash tools/all_configs.sh > gpurun_out/all_configs.txt
cd "${GRAFT_REPO_ROOT:-/root/repo}"
run() {
    echo "== bench.py $*"
    python3 bench.py --no-cpu-baseline --extras pipelined --steps 10 --warmup 3 "$@" 2>/dev/null | tail -1 | python3 -c "
import sys, json
d = json.loads(sys.stdin.readline()); r = d['roofline']; c = d['config']
print(f\"{d['value']:.1f} Mrays/s  {d['ms_per_step']:.4f} ms/frame  integrate {r['avg_launch_ms']:.4f} ms  prologue/epilogue {r['other_kernels_ms']['prologue']:.3f}/{r['other_kernels_ms']['epilogue']:.3f} ms  \"
      f\"issue frac {r['frac']} (as-written {r['algorithmic_as_written']['of_peak']})  steps/ray {c['mean_rk4_steps_per_ray']}  escaped/captured/invalid {c['escaped']}/{c['captured']}/{c['invalid']}\"
      + (f\"  pipelined x{d['pipelined']['frames_in_flight']}: {d['pipelined']['value']:.1f} Mrays/s\" if 'pipelined' in d else ''))"
}
run --metric schwarzschild --size 1024
run --size 2048
run --size 2048 --schedule queue
run --size 2048 --integrator dp45 --precision 64
run --size 4096
run --size 4096 --integrator dp45 --precision 64
run --size 4096 --r-obs 100 --background
run --size 8192 --a 0.99
run --size 16384 --steps 2 --warmup 1
run --size 32768 --steps 1 --warmup 1
