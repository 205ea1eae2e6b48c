#!/usr/bin/env python3
"""bench.py -- north-star benchmark: Mrays/s of a 4096x4096 Kerr (a = 0.9) shadow render,
fixed-step RK4 in float32, on N MI355X of one node.

  python bench.py --gpus 1 --steps 20 --warmup 3
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one whole frame: every rank renders its block-cyclic share of the pixel rows
(prologue + integrate + epilogue kernels of libltrace_hip.so, launched on torch's current
stream), then the RGBA8 framebuffer is gathered to rank 0 over RCCL and un-permuted there.
The frame is fixed as N grows, so scaling is "strong".  Everything the timed region reads is
already resident in HBM; no per-ray input exists (pixel -> ray happens in the prologue kernel).

Rank 0 prints ONE JSON line.  `roofline` prices the integrate kernel against the FP32 VALU
peak with the reference's as-written flop counts; `cpu_baseline` is the oracle
(CPU port of the reference algorithm, float64, OpenMP) timed on this host on a strided
subsample of the same frame.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "light-path-tracer_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

import ltrace  # noqa: E402
import sharding  # noqa: E402

# As-written flop counts of the reference
F_RK4_STEP = 4 * 188 + 80     # 832 per RK4 step   (metrics.py:221-323)
F_DP45_ATTEMPT = 6 * 188 + 320  # 1448 per DP45 step attempt (metrics.py:454-564); + 188 once for the FSAL seed
F_KERR_FIXED = 95 + 72        # 167 per ray        (metrics.py:148-218, :363-416)
F_SCHW_STEP = 54
F_SCHW_FIXED = 45
PEAK_FP32_VALU_TFLOPS = 157.3  # MI355X_MICROARCH.md, chip-level parameters
PEAK_FP64_VALU_TFLOPS = 78.6   # float64 vector rate = half the float32 one


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--size", type=int, default=4096, help="frame is size x size")
    ap.add_argument("--a", type=float, default=0.9)
    ap.add_argument("--r-obs", type=float, default=50.0)
    ap.add_argument("--metric", choices=["kerr", "schwarzschild"], default="kerr")
    ap.add_argument("--schedule", choices=["direct", "queue"], default=os.environ.get("LT_SCHEDULE", "direct"))
    ap.add_argument("--precision", type=int, default=32)
    ap.add_argument("--integrator", choices=["rk4", "dp45"], default="rk4")
    ap.add_argument("--row-block", type=int, default=16)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="target wall time of the CPU baseline leg")
    ap.add_argument("--background", action="store_true", help="image_lens workload: lens a synthetic background")
    return ap.parse_args()


def cpu_baseline(args, fov):
    """Oracle (CPU port of the reference's RK4 tracer, float64, all cores) on a strided subsample of
    the benchmark frame: pixel (k*i, k*j) of the size^2 frame is pixel (i, j) of the (size/k)^2 frame
    of the same camera."""
    from oracle import oracle
    kind = args.metric
    t0 = time.perf_counter()
    oracle.lookup(kind, 1.0, args.a, args.r_obs, 256, 256, fov, fov, integrator=args.integrator)
    dt = time.perf_counter() - t0          # calibration (includes thread spin-up)
    rate = 256 * 256 / dt
    stride = 2
    while stride < args.size // 256 and (args.size // stride) ** 2 / rate > args.cpu_seconds:
        stride *= 2
    n = args.size // stride
    t0 = time.perf_counter()
    r = oracle.lookup(kind, 1.0, args.a, args.r_obs, n, n, fov, fov, integrator=args.integrator)
    dt = time.perf_counter() - t0
    return {"value": round(n * n / dt / 1e6, 4), "unit": "Mrays/s", "cores": oracle.num_threads(),
            "kind": "port",
            "sample": f"{n}x{n} rays = every {stride}th pixel (x and y) of the {args.size}x{args.size} frame, "
                      f"oracle {args.integrator} float64 + OpenMP, {dt:.1f} s",
            "mean_rhs_evals_per_ray": round(float(r["evals"].mean()), 1)}


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", device_id=dev)

    size = args.size
    fov = float(np.radians(40.0))
    cam = ltrace.Camera(size, size, fov, fov, 0.0, 0.0, args.r_obs, np.pi / 2)
    met = ltrace.Metric(ltrace.METRIC_KERR if args.metric == "kerr" else ltrace.METRIC_SCHWARZSCHILD, 0, 1.0,
                        args.a if args.metric == "kerr" else 0.0)
    rb = args.row_block
    rows_max = max(ltrace.local_rows(size, rb, world, p) for p in range(world))

    # device buffers (torch owns the memory; the library only sees raw pointers)
    fg = sharding.FrameGather(size, size, 4, torch.uint8, dev, rb, world, rank)   # RGBA8 framebuffer
    d_rgba = fg.local
    d_fa = torch.empty((rows_max, size), dtype=torch.float32, device=dev)
    d_w = torch.empty((rows_max, size), dtype=torch.int16, device=dev)
    d_stats = torch.zeros(ltrace.STAT_WORDS, dtype=torch.int64, device=dev)
    d_bg = None
    if args.background:
        g = torch.Generator(device="cpu").manual_seed(0)
        d_bg = (torch.randint(0, 256, (size, size, 3), generator=g, dtype=torch.uint8).to(torch.float32) / 255.0).to(dev)

    stream = torch.cuda.current_stream(dev)
    opts = ltrace.default_opts(integrator=args.integrator, precision=args.precision, schedule=args.schedule,
                               row_block=rb, n_parts=world, part=rank, timing=1)
    opts.stream = stream.cuda_stream

    def step():
        ltrace.render_dev(cam, met, opts, d_bg=d_bg.data_ptr() if d_bg is not None else 0, bg_channels=3,
                          d_fa=d_fa.data_ptr(), d_w=d_w.data_ptr(), d_rgba=d_rgba.data_ptr(),
                          d_stats=d_stats.data_ptr())
        return fg.gather(stream.cuda_stream)   # N > 1: RCCL gather to rank 0 + row un-permute there

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        step()
    fence()
    ltrace.timing_collect()            # drop warm-up events
    d_stats.zero_()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    tm = ltrace.timing_collect()

    t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    counters = d_stats.clone()
    kern = torch.tensor([tm["prologue_ms"], tm["integrate_ms"], tm["epilogue_ms"]], dtype=torch.float64, device=dev)
    kern_max = kern.clone()
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(counters, op=dist.ReduceOp.SUM)
        dist.all_reduce(kern_max, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    c = [int(x) for x in counters.tolist()]
    steps = max(args.steps, 1)
    rays_per_frame = c[ltrace.STAT_RAYS] // steps
    rk_steps = c[ltrace.STAT_STEPS] / steps
    if args.metric == "kerr":
        if args.integrator == "dp45":
            flops_frame = rk_steps * F_DP45_ATTEMPT + rays_per_frame * (F_KERR_FIXED + 188)
        else:
            flops_frame = rk_steps * F_RK4_STEP + rays_per_frame * F_KERR_FIXED
    else:
        flops_frame = rk_steps * F_SCHW_STEP + rays_per_frame * F_SCHW_FIXED
    ms_per_step = elapsed / steps * 1e3
    value = rays_per_frame / (elapsed / steps) / 1e6

    if rank == 0:
        # HBM bytes per launch of the integrate kernel: PMC counters cannot be read from inside this
        # process, so the figure comes from the committed rocprofv3 pass for this exact workload
        spin = args.a if args.metric == "kerr" else 0.0
        workload = f"{args.metric}_a{spin}_shadow_{size}x{size}_r{args.r_obs:g}_{args.integrator}"
        traffic = None
        try:
            with open(os.path.join(ROOT, "profiles", "hbm_traffic.json")) as f:
                rec = json.load(f).get(workload)
            if rec and world == 1 and args.precision == 32 and not args.background:
                traffic = rec.get(args.schedule)
        except OSError:
            pass
        integ_ms = float(kern_max[1].item()) / steps          # slowest rank's average integrate-kernel time
        achieved = (flops_frame / world) / (integ_ms * 1e-3) / 1e12 if integ_ms > 0 else 0.0
        peak = PEAK_FP32_VALU_TFLOPS if args.precision == 32 else PEAK_FP64_VALU_TFLOPS
        out = {
            "metric": "Mrays/s", "value": round(value, 2), "unit": "Mrays/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f32" if args.precision == 32 else "f64", "data": "synthetic",
            "config": {"workload": workload
                                   + ("_lensed_background" if args.background else ""),
                       "rays_per_frame": rays_per_frame, "schedule": args.schedule,
                       "row_partition": f"block-cyclic {rb} rows x {world}", "gather": "rccl" if world > 1 else "none",
                       "mean_rk4_steps_per_ray": round(rk_steps / max(rays_per_frame, 1), 2),
                       "escaped": c[ltrace.STAT_ESCAPED] // steps, "captured": c[ltrace.STAT_CAPTURED] // steps,
                       "invalid": c[ltrace.STAT_INVALID] // steps},
            "roofline": {"bound": "valu_fp32" if args.precision == 32 else "valu_fp64", "kernel": f"k_kerr_{args.schedule}<{args.integrator}>" if args.metric == "kerr" else "k_schw_rk4_direct",
                         "achieved": round(achieved, 2), "peak": peak, "unit": "TFLOP/s",
                         "frac": round(achieved / peak, 4), "traffic": traffic,
                         "algorithmic_flops_per_launch": int(flops_frame / world),
                         "avg_launch_ms": round(integ_ms, 4),
                         "other_kernels_ms": {"prologue": round(float(kern_max[0].item()) / steps, 4),
                                              "epilogue": round(float(kern_max[2].item()) / steps, 4)}},
        }
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(args, fov)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
