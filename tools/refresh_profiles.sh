#!/bin/bash
# Regenerate everything under profiles/ in one go.
#   on the GPU box (through gpurun):   ROUND=02 bash tools/refresh_profiles.sh measure      (writes gpurun_out/)
#   afterwards, in the build container: ROUND=02 bash tools/refresh_profiles.sh collect      (copies into profiles/rNN_*)
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
R=r${ROUND:-02}
G=gpurun_out
if [ "${1:-}" = measure ]; then
    python3 tools/pmc_counts.py > $G/pmc_counts.log 2>&1                     # VALU / HBM counters per workload, keyed by build id
    echo pmc_counts done; tail -3 $G/pmc_counts.log
    bash tools/prof.sh direct > $G/prof_direct.log 2>&1                      # kernel trace + SQ counters, north-star frame
    bash tools/prof.sh queue --schedule queue > $G/prof_queue.log 2>&1
    bash tools/prof.sh dp45 --integrator dp45 --precision 64 > $G/prof_dp45.log 2>&1
    bash tools/prof.sh imagelens --r-obs 100 --background > $G/prof_imagelens.log 2>&1
    echo prof done
    mkdir -p $G/pc_tmp && cp $G/pmc_counts/valu_counts.json $G/pmc_counts/hbm_traffic.json profiles/ 2>/dev/null  # so that the lines below carry frac / traffic
    python3 bench.py > $G/bench_direct.json 2>$G/bench_direct.err
    python3 bench.py --schedule queue --no-cpu-baseline > $G/bench_queue.json 2>/dev/null
    python3 bench.py --integrator dp45 --precision 64 --no-cpu-baseline > $G/bench_dp45.json 2>/dev/null
    python3 bench.py --integrator dp45_exact --precision 64 --no-cpu-baseline --no-extras --steps 5 > $G/bench_dp45_exact.json 2>/dev/null
    python3 bench.py --r-obs 100 --background --no-cpu-baseline > $G/bench_imagelens.json 2>/dev/null
    python3 bench.py --size 2048 --no-cpu-baseline > $G/bench_2048.json 2>/dev/null
    echo bench done
    bash tools/all_configs.sh > $G/all_configs.txt 2>&1
    bash tools/pipeline_bench.sh > $G/pipeline.txt 2>&1
    python3 tools/part_bench.py 4096 rk4 > $G/part_bench_rk4.log 2>&1
    python3 tools/part_bench.py 4096 dp45 > $G/part_bench_dp45.log 2>&1
    python3 tools/part_bench.py 8192 rk4 5 0.99 > $G/part_bench_8192.log 2>&1
    python3 tools/balance_bench.py 4096 > $G/balance_bench.log 2>&1
    for p in 25 205 50 75 0; do python3 bench.py --size 4096 --no-cpu-baseline --no-extras --steps 10 --emulate-parts 256 --emulate-part $p 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.readline()); print('row block $p alone (16 rows x 4096): integrate', d['roofline']['avg_launch_ms'], 'ms')"; done > $G/single_blocks.log 2>&1
    python3 tools/long_ray_pace.py > $G/long_ray_pace.log 2>&1
    python3 tools/long_ray_pace.py 2048 > $G/long_ray_pace_2048.log 2>&1
    python3 tools/lone_step.py > $G/lone.log 2>&1
    LT_STAMPS_FILE=/tmp/lt_stamps_lanes.bin LT_D_LONG=2000000000 python3 tools/lone_pace_by_lanes.py > $G/lone_lanes_off.log 2>&1
    LT_STAMPS_FILE=/tmp/lt_stamps_lanes.bin python3 tools/lone_pace_by_lanes.py > $G/lone_lanes_on.log 2>&1
    python3 tools/e2e_frame.py > $G/e2e_frame.log 2>&1
    bash tools/pmc_once.sh valubusy "VALUBusy" > $G/valubusy.log 2>&1
    echo chain done
    bash tools/prof_dense.sh > /dev/null 2>&1
    python3 tools/dense_bench.py 65536 512 2>/dev/null > $G/dense_bench.log
    python3 tools/dense_bench.py 4194304 224 2>/dev/null >> $G/dense_bench.log
    python3 tools/dense_lane_stats.py > $G/dense_lane_stats.log 2>&1
    python3 -m pytest tests/test_gpu_dense.py -m gpu -x -q -s -k batch_matches 2>&1 | grep oracle > $G/dense_oracle.log
    tail -1 $G/bench_direct.json | cut -c1-300
elif [ "${1:-}" = collect ]; then
    P=profiles
    cp $G/pmc_counts/valu_counts.json $G/pmc_counts/hbm_traffic.json $P/
    { echo "# $R: executed VALU instructions and HBM bytes of the integrate kernel per launch, per workload (tools/pmc_counts.py;"
      echo "# rocprofv3 --pmc, separate passes for SQ / FETCH_SIZE / WRITE_SIZE / TCC; build id $(python3 -c "import json;print(json.load(open('$P/valu_counts.json'))['build_id'])"))"
      cat $G/pmc_counts/summary.txt; } > $P/${R}_pmc_counts.txt
    for t in direct queue imagelens; do
        cp $G/prof_$t/summary.txt $P/${R}_${t}_rocprofv3_summary.txt
        cp "$(ls -t $G/prof_$t/trace/*/*_kernel_stats.csv | head -1)" $P/${R}_${t}_kernel_stats.csv
    done
    for t in direct queue dp45 dp45_exact imagelens 2048; do tail -1 $G/bench_$t.json > $P/${R}_bench_$t.json; done
    { echo "# $R: 4096x4096 Kerr a=0.9 with the reference's production integrator (DP45, float64): tools/prof.sh dp45 --integrator dp45 --precision 64"
      grep -v "at::native\|rocclr" $G/prof_dp45/summary.txt | grep -A14 "kernel  \|^  lt::k_kerr_direct" | grep -v "^--"; } > $P/${R}_dp45_rocprofv3_summary.txt
    { echo "# $R: every BASELINE.json config shape that fits one MI355X, plus two frames beyond them (tools/all_configs.sh)"; cat $G/all_configs.txt; } > $P/${R}_all_configs_one_gpu.txt
    { echo "# $R: frames in flight (bench.py --frames-in-flight F): frame i on stream i % F with its own buffers, so the tail of a frame"
      echo "# overlaps the bulk of the next; whole frame on one GPU, and what ONE rank of an N-GPU run renders (--emulate-parts N). tools/pipeline_bench.sh"
      cat $G/pipeline.txt; } > $P/${R}_frames_in_flight.txt
    { echo "# $R: the serial chain that bounds small launches and strong scaling (4096x4096 Kerr a=0.9, RK4 float32)"
      echo "# tools/long_ray_pace.py: the longest rays of the frame, each traced ALONE on the chip (one wavefront)"; grep -v amdgpu $G/long_ray_pace.log
      echo "# tools/long_ray_pace.py 2048: the same for the 2048x2048 frame (config 3)"; grep -v amdgpu $G/long_ray_pace_2048.log
      echo "# tools/lone_step.py: bare RK4 step (lt_rk4_step_probe, probe build), cycles per wave-step by resident waves per SIMD"; grep -v amdgpu $G/lone.log
      echo "# tools/lone_pace_by_lanes.py with LT_D_LONG=2000000000 (no wave ever uses ghost lanes): the pace of a lone wavefront by the number of its lanes still enabled"; grep -v amdgpu $G/lone_lanes_off.log
      echo "# tools/lone_pace_by_lanes.py, defaults (ghost lanes after 1024 iterations)"; grep -v amdgpu $G/lone_lanes_on.log
      echo "# tools/part_bench.py 4096 rk4: one rank of an N-GPU run under benchmark conditions"; grep n_parts $G/part_bench_rk4.log
      echo "# tools/part_bench.py 4096 dp45 (float64)"; grep n_parts $G/part_bench_dp45.log
      echo "# tools/part_bench.py 8192 rk4 5 0.99 (config 5: Kerr a = 0.99, 8192 x 8192, rows sharded across 8 GPUs)"; grep n_parts $G/part_bench_8192.log
      echo "# single 16-row blocks rendered alone (bench.py --emulate-parts 256 --emulate-part b): blocks 25, 205, 50, 75 hold the four longest rays, block 0 none"; cat $G/single_blocks.log
      echo "# tools/balance_bench.py 4096: cost-weighted row-block assignment (sharding.balance_blocks, lt_opts.block_owner) against block-cyclic, every rank emulated on one GPU"; grep -v amdgpu $G/balance_bench.log; } > $P/${R}_long_ray_chain.txt
    { echo "# $R: host-pointer lt_render, 4096x4096, RGBA8 destination only (tools/e2e_frame.py): what python image_lens.py pays per frame"
      grep -v amdgpu $G/e2e_frame.log; } > $P/${R}_end_to_end_frame.txt
    { echo "# $R: rocprofv3 --pmc VALUBusy (derived metric) over bench.py --steps 3 (tools/pmc_once.sh valubusy VALUBusy)"
      grep -A1 "k_kerr_direct\|k_epilogue" $G/valubusy.log | grep -v "^--"; } > $P/${R}_valu_busy.txt
    { echo "# $R: batched dense trajectories (lt_integrate_dense_dev), tools/prof_dense.sh + tools/dense_bench.py + tools/dense_lane_stats.py"
      grep -A3 "^== kernel stats" $G/prof_dense/summary.txt | head -3; grep -A12 "^  lt::k_dense_tracks" $G/prof_dense/summary.txt | head -13
      cat $G/prof_dense/bench.json $G/dense_bench.log; grep -v amdgpu $G/dense_lane_stats.log; cat $G/dense_oracle.log; } > $P/${R}_dense_tracks.txt
    ls -la $P | tail -30
else
    echo "usage: $0 measure|collect"; exit 2
fi
