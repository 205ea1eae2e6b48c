#!/bin/bash
# Regenerate everything under profiles/ in one go.
#   on the GPU box (through gpurun):   ROUND=03 bash tools/refresh_profiles.sh measure      (writes gpurun_out/)
#   afterwards, in the build container: ROUND=03 bash tools/refresh_profiles.sh collect      (copies into profiles/rNN_*)
# `measure quick`: only what the driver's bench line reads (PMC counts of the two regions) + the two profiles it is checked against.
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
R=r${ROUND:-03}
G=gpurun_out
if [ "${1:-}" = measure ]; then
    QUICK=${2:-}
    python3 tools/pmc_counts.py ${QUICK:+--quick} > $G/pmc_counts.log 2>&1              # VALU / HBM counters per workload, keyed by build id
    echo pmc_counts done; tail -3 $G/pmc_counts.log
    mkdir -p $G/pc_tmp && cp $G/pmc_counts/valu_counts.json $G/pmc_counts/hbm_traffic.json profiles/ 2>/dev/null  # so that the lines below carry frac / traffic
    bash tools/prof.sh direct > $G/prof_direct.log 2>&1                      # plain / traced / plain + SQ counters, north-star frame
    PROF_PMC=0 bash tools/prof.sh dp45_exact --integrator dp45_exact --precision 64 > $G/prof_dp45_exact.log 2>&1
    python3 bench.py --steps 20 --warmup 5 > $G/bench_default.json 2>$G/bench_default.err   # the driver's command line
    echo "driver line done"; tail -c 300 $G/bench_default.json
    [ -n "$QUICK" ] && exit 0
    bash tools/prof.sh queue --schedule queue > $G/prof_queue.log 2>&1
    PROF_PMC=0 bash tools/prof.sh dp45 --integrator dp45 --precision 64 > $G/prof_dp45.log 2>&1
    bash tools/prof.sh imagelens --r-obs 100 --background > $G/prof_imagelens.log 2>&1
    echo prof done
    python3 bench.py --schedule queue --no-cpu-baseline --extras pipelined > $G/bench_queue.json 2>/dev/null
    python3 bench.py --integrator dp45 --precision 64 --no-cpu-baseline --extras pipelined > $G/bench_dp45.json 2>/dev/null
    python3 bench.py --r-obs 100 --background --no-cpu-baseline --extras pipelined,chain > $G/bench_imagelens.json 2>/dev/null
    python3 bench.py --size 2048 --no-cpu-baseline > $G/bench_2048.json 2>/dev/null
    echo bench done
    bash tools/all_configs.sh > $G/all_configs.txt 2>&1
    bash tools/pipeline_bench.sh > $G/pipeline.txt 2>&1
    python3 tools/part_bench.py 8192 rk4 5 0.99 > $G/part_bench_8192.log 2>&1
    python3 tools/balance_bench.py 4096 > $G/balance_bench.log 2>&1
    python3 tools/long_ray_pace.py > $G/long_ray_pace.log 2>&1
    python3 tools/long_ray_pace.py 2048 > $G/long_ray_pace_2048.log 2>&1
    python3 tools/e2e_frame.py > $G/e2e_frame.log 2>&1
    python3 tools/parity_stats.py > $G/parity_stats.txt 2>/dev/null
    echo chain done
    bash tools/prof_dense.sh binned 4194304 224 0.9 1 > /dev/null 2>&1
    bash tools/prof_dense.sh caller 4194304 224 0.9 -1 > /dev/null 2>&1
    bash tools/pmc_dense.sh 4194304 224 > /dev/null 2>&1
    python3 tools/merge_dense_records.py
    for b in -1 0; do for n in 65536 1048576 4194304; do python3 tools/dense_bench.py $n 224 0.9 $b 2>/dev/null; done; done > $G/dense_bench.log
    python3 tools/dense_lane_stats.py > $G/dense_lane_stats.log 2>&1
    python3 -m pytest tests/test_gpu_dense.py -m gpu -q -s -k "batch_matches or length_predictor" 2>&1 | grep "oracle\|lane-utilisation" > $G/dense_oracle.log
    tail -1 $G/bench_default.json | cut -c1-300
elif [ "${1:-}" = collect ]; then
    P=profiles
    cp $G/pmc_counts/valu_counts.json $G/pmc_counts/hbm_traffic.json $P/
    { echo "# $R: executed VALU instructions and HBM bytes of the integrate kernel per launch, per workload (tools/pmc_counts.py;"
      echo "# rocprofv3 --pmc, separate passes for SQ / FETCH_SIZE / WRITE_SIZE / TCC; build id $(python3 -c "import json;print(json.load(open('$P/valu_counts.json'))['build_id'])"))"
      cat $G/pmc_counts/summary.txt; } > $P/${R}_pmc_counts.txt
    for t in direct dp45_exact queue dp45 imagelens; do
        [ -f $G/prof_$t/summary.txt ] || continue
        cp $G/prof_$t/summary.txt $P/${R}_${t}_rocprofv3_summary.txt
        cp "$(ls -t $G/prof_$t/trace/*/*_kernel_stats.csv | head -1)" $P/${R}_${t}_kernel_stats.csv
    done
    for t in default queue dp45 imagelens 2048; do [ -s $G/bench_$t.json ] && tail -1 $G/bench_$t.json > $P/${R}_bench_$t.json; done
    [ -s $G/all_configs.txt ] && { echo "# $R: every BASELINE.json config shape that fits one MI355X, plus two frames beyond them (tools/all_configs.sh)"; cat $G/all_configs.txt; } > $P/${R}_all_configs_one_gpu.txt
    [ -s $G/pipeline.txt ] && { echo "# $R: frames in flight (bench.py --frames-in-flight F): frame i on stream i % F with its own buffers, so the tail of a frame"
      echo "# overlaps the bulk of the next; whole frame on one GPU, and what ONE rank of an N-GPU run renders (--emulate-parts N). tools/pipeline_bench.sh"
      cat $G/pipeline.txt; } > $P/${R}_frames_in_flight.txt
    [ -s $G/long_ray_pace.log ] && { echo "# $R: the serial chain that bounds small launches and strong scaling (4096x4096 Kerr a=0.9, RK4 float32)"
      echo "# tools/long_ray_pace.py: the longest rays of the frame, each traced ALONE on the chip (one wavefront)"; grep -v amdgpu $G/long_ray_pace.log
      echo "# tools/long_ray_pace.py 2048: the same for the 2048x2048 frame (config 3)"; grep -v amdgpu $G/long_ray_pace_2048.log
      echo "# per-rank frame times of 2 / 4 / 8 partitions of the 4096^2 frame: the projected_ranks block of ${R}_bench_default.json (RK4 float32 and DP45 float64)"
      python3 -c "
import json; d = json.load(open('$P/${R}_bench_default.json'))
for k, v in d.get('projected_ranks', {}).items():
    for n in ('2', '4', '8'): print(k, 'n_parts=' + n, v[n])"
      echo "# tools/part_bench.py 8192 rk4 5 0.99 (config 5: Kerr a = 0.99, 8192 x 8192, rows sharded across 8 GPUs)"; grep n_parts $G/part_bench_8192.log
      echo "# tools/balance_bench.py 4096: cost-weighted row-block assignment (sharding.balance_blocks, lt_opts.block_owner) against block-cyclic, every rank emulated on one GPU"; grep -v amdgpu $G/balance_bench.log; } > $P/${R}_long_ray_chain.txt
    [ -s $G/e2e_frame.log ] && { echo "# $R: host-pointer lt_render, 4096x4096, RGBA8 destination only (tools/e2e_frame.py): what python image_lens.py pays per frame"
      grep -v amdgpu $G/e2e_frame.log; } > $P/${R}_end_to_end_frame.txt
    [ -s $G/parity_stats.txt ] && { echo "# $R: GPU batch tracers against every golden per-ray fixture (the reference's own outputs), tools/parity_stats.py on the MI355X box"; cat $G/parity_stats.txt; } > $P/${R}_parity_stats.txt
    [ -f $G/prof_dense_binned/summary.txt ] && { echo "# $R: batched dense trajectories (lt_integrate_dense_dev), 4 M tracks: tools/prof_dense.sh (length-binned launch, then caller order), tools/pmc_dense.sh, tools/dense_bench.py, tools/dense_lane_stats.py"
      echo "## length-binned launch (predictor pass + windowed counting sort + tracks longest-first per window)"; grep -v "^  lt::k_dense_window\|^  lt::k_dense_predict" $G/prof_dense_binned/summary.txt | grep -B2 -A12 "kernel trace\|^  lt::k_dense_tracks" | grep -v "^--"
      echo "## caller order"; grep -A4 "kernel trace" $G/prof_dense_caller/summary.txt; grep -A12 "^  lt::k_dense_tracks" $G/prof_dense_caller/summary.txt
      echo "## memory-pipe counters of k_dense_tracks, caller order against length-binned (tools/pmc_dense.sh)"; cat $G/pmc_dense/summary.txt
      echo "## throughput by batch size: length_binning -1 (caller order) and 0 (automatic)"; cat $G/dense_bench.log
      grep -v amdgpu $G/dense_lane_stats.log; cat $G/dense_oracle.log; } > $P/${R}_dense_tracks.txt
    python3 tools/merge_dense_records.py
    ls -la $P | grep $R
else
    echo "usage: $0 measure [quick]|collect"; exit 2
fi
