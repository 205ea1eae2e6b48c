#!/bin/bash
# Regenerate everything under profiles/ in one go.
#   on the GPU box (through gpurun):   bash tools/refresh_profiles.sh measure      (writes gpurun_out/)
#   afterwards, in the build container: bash tools/refresh_profiles.sh collect      (copies into profiles/rNN_*)
# ROUND defaults to 01.
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
R=r${ROUND:-01}
G=gpurun_out
if [ "${1:-}" = measure ]; then
    bash tools/prof.sh direct > $G/prof_direct.log 2>&1
    bash tools/prof.sh queue --schedule queue > $G/prof_queue.log 2>&1
    bash tools/prof.sh dp45 --integrator dp45 --precision 64 > $G/prof_dp45.log 2>&1
    python3 bench.py > $G/bench_direct.json 2>/dev/null
    python3 bench.py --schedule queue --no-cpu-baseline > $G/bench_queue.json 2>/dev/null
    python3 bench.py --integrator dp45 --precision 64 --no-cpu-baseline > $G/bench_dp45.json 2>/dev/null
    bash tools/all_configs.sh > $G/all_configs.txt 2>&1
    python3 tools/part_bench.py 4096 rk4 > $G/part_bench_rk4.log 2>&1
    python3 tools/part_bench.py 4096 dp45 > $G/part_bench_dp45.log 2>&1
    python3 tools/long_ray_pace.py > $G/long_ray_pace.log 2>&1
    python3 tools/lone_step.py > $G/lone.log 2>&1
    python3 tools/pace_vs_load.py > $G/pace_load.log 2>&1
    bash tools/pmc_once.sh valubusy "VALUBusy" > $G/valubusy.log 2>&1
    bash tools/prof_dense.sh > /dev/null 2>&1
    python3 tools/dense_bench.py 65536 512 2>/dev/null > $G/dense_bench.log
    python3 tools/dense_bench.py 4194304 224 2>/dev/null >> $G/dense_bench.log
    python3 -m pytest tests/test_gpu_dense.py -m gpu -x -q -s -k batch_matches 2>&1 | grep oracle > $G/dense_oracle.log
    tail -1 $G/bench_direct.json | cut -c1-200
elif [ "${1:-}" = collect ]; then
    P=profiles
    for t in direct queue; do
        cp $G/prof_$t/summary.txt $P/${R}_${t}_rocprofv3_summary.txt
        cp "$(ls -t $G/prof_$t/trace/*/*_kernel_stats.csv | head -1)" $P/${R}_${t}_kernel_stats.csv
        tail -1 $G/bench_$t.json > $P/${R}_bench_$t.json
    done
    tail -1 $G/bench_dp45.json > $P/${R}_bench_dp45_f64.json
    { echo "# $R: 4096x4096 Kerr a=0.9 with the reference's production integrator (DP45, float64): tools/prof.sh dp45 --integrator dp45 --precision 64"
      grep -v "at::native\|rocclr" $G/prof_dp45/summary.txt | grep -A12 "kernel  \|^  lt::k_kerr_direct" | grep -v "^--"; } > $P/${R}_dp45_rocprofv3_summary.txt
    { echo "# $R: every BASELINE.json config shape that fits one MI355X, plus two frames beyond them (tools/all_configs.sh)"; cat $G/all_configs.txt; } > $P/${R}_all_configs_one_gpu.txt
    { echo "# $R: the serial chain that bounds small launches and strong scaling (4096x4096 Kerr a=0.9, RK4 float32)"
      echo "# tools/long_ray_pace.py: the longest rays of the frame, each traced ALONE on the chip (one wavefront)"; grep -v amdgpu $G/long_ray_pace.log
      echo "# tools/lone_step.py: bare RK4 step (lt_rk4_step_probe), cycles per wave-step by resident waves per SIMD"; cat $G/lone.log
      echo "# tools/part_bench.py 4096 rk4: one rank of an N-GPU run under benchmark conditions"; grep n_parts $G/part_bench_rk4.log
      echo "# tools/part_bench.py 4096 dp45 (float64)"; grep n_parts $G/part_bench_dp45.log
      echo "# tools/pace_vs_load.py: one long ray replicated into identical wavefronts (times include ~0.2 ms of call overhead)"; grep identical $G/pace_load.log; } > $P/${R}_long_ray_chain.txt
    { echo "# $R: rocprofv3 --pmc VALUBusy (derived metric) over bench.py --steps 3 (tools/pmc_once.sh valubusy VALUBusy)"
      grep -A1 "k_kerr_direct\|k_epilogue" $G/valubusy.log | grep -v "^--"; } > $P/${R}_valu_busy.txt
    { echo "# $R: batched dense trajectories (lt_integrate_dense_dev), tools/prof_dense.sh + tools/dense_bench.py"
      grep -A3 "^== kernel stats" $G/prof_dense/summary.txt | head -3; grep -A6 "^  lt::k_dense_tracks" $G/prof_dense/summary.txt | head -7
      cat $G/prof_dense/bench.json $G/dense_bench.log $G/dense_oracle.log; } > $P/${R}_dense_tracks.txt
    ls -la $P | tail -20
else
    echo "usage: $0 measure|collect"; exit 2
fi
