#!/usr/bin/env python3
"""bench.py -- north-star benchmark: Mrays/s of a 4096x4096 Kerr (a = 0.9) shadow render,
fixed-step RK4 in float32, on N MI355X of one node.

  python bench.py --gpus N --steps K --warmup W          (starts its own N ranks when N > 1)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one whole frame: every rank renders its block-cyclic share of the pixel rows
(prologue + integrate + epilogue kernels of libltrace_hip.so, launched on torch's current
stream), then the RGBA8 framebuffer is gathered to rank 0 over RCCL and un-permuted there.
The frame is fixed as N grows, so scaling is "strong".  Everything the timed region reads is
already resident in HBM; no per-ray input exists (pixel -> ray happens in the prologue kernel).

Ranks.  `--gpus N` means N processes, one per GPU.  Started under torchrun (RANK / WORLD_SIZE set)
this file is one of them and WORLD_SIZE must equal N.  Started plainly with N > 1 it is the
launcher: BEFORE anything touches the GPU it checks that the node has N devices (exit 2 otherwise)
and starts N copies of itself as child processes with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*
set, waits for them and exits with their worst code.  It never replaces a process that has
initialised the GPU.  A rank whose process group does not have N members exits non-zero.

Rank 0 prints ONE JSON line.  Its contract fields are the headline region (fixed-step RK4, float32: the
integrator BASELINE.json's north star names).  `production_path` is a second timed region of the same
frame, same K and W, with the integrator the reference itself runs (`python image_lens.py --a 0.9` ->
metrics.py:1128-1132 -> DP45 float64, metrics.py:419-567; the product's plugin default), with its own
roofline and its own CPU baseline.  `projected_ranks` (N = 1): what each rank of a 2 / 4 / 8 GPU run
renders, one rank at a time on this GPU -- a projection, labelled as one.
  roofline      the integrate kernel against the FP32 VALU ISSUE peak, from the work it executed:
                wave-instructions per launch (rocprofv3 SQ_INSTS_VALU for this workload and this
                build, profiles/valu_counts.json, scaled by the loop iterations the kernel counted
                in THIS run) x 128 lane-flops per issue slot / the kernel's HIP-event time.  One
                VALU instruction occupies a SIMD for 2 cycles (MI355X_MICROARCH.md), so the peak is
                1024 SIMDs x 2.4 GHz / 2 = 1.2288e12 wave-instructions/s = 157.3 TFLOP/s of FMAs and
                frac <= 1 by construction.  The reference's as-written flop count, which
                the kernel does not execute, is reported beside it as `algorithmic_as_written`.
  cpu_baseline  the oracle (CPU port of the reference algorithm, float64, OpenMP) compiled
                -O3 -march=native on this host, on a strided subsample of the same frame.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "light-path-tracer_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

# As-written flop counts of the reference
F_RK4_STEP = 4 * 188 + 80     # 832 per RK4 step   (metrics.py:221-323)
F_DP45_ATTEMPT = 6 * 188 + 320  # 1448 per DP45 step attempt (metrics.py:454-564); + 188 once for the FSAL seed
F_KERR_FIXED = 95 + 72        # 167 per ray        (metrics.py:148-218, :363-416)
F_SCHW_STEP = 54
F_SCHW_FIXED = 45
PEAK_FP32_VALU_TFLOPS = 157.3  # MI355X_MICROARCH.md, chip-level parameters
PEAK_FP64_VALU_TFLOPS = 78.6   # float64 vector rate = half the float32 one
N_SIMD = 256 * 4               # CUs x SIMDs
NOMINAL_CLOCK_HZ = 2.4e9
CYCLES_PER_VALU = {32: 2.0, 64: 4.0}   # issue cycles of one wave64 FMA-class instruction on a SIMD-32


def parse(argv=None):
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
    ap.add_argument("--integrator", choices=["rk4", "dp45", "dp45_exact"], default="rk4")
    ap.add_argument("--row-block", type=int, default=16)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true",
                    help="only the headline region: no pipelined region, no production-path region, no end-to-end host-pointer "
                         "frame, no longest-ray chain, no per-rank projection (what the profiling scripts pass)")
    ap.add_argument("--no-production-path", action="store_true",
                    help="skip the second timed region (the same frame with the reference's production integrator, DP45 float64)")
    ap.add_argument("--extras", default="pipelined,production,chain,e2e,projection",
                    help="which of the extras to run (comma list; --no-extras = none): pipelined (the K frames again, 3 in flight), "
                         "production (second timed region, DP45 float64), chain (longest ray alone), e2e (host-pointer frame), "
                         "projection (each rank of 2 / 4 / 8 alone on this GPU)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="target wall time of each CPU baseline leg")
    ap.add_argument("--background", action="store_true", help="image_lens workload: lens a synthetic background")
    ap.add_argument("--frames-in-flight", type=int, default=1,
                    help="1: frames strictly one after the other on one stream (the contract's per-launch roofline). "
                         "F > 1: frame i runs on stream i %% F with its own buffers, so the tail of one frame (a few "
                         "lone wavefronts finishing the longest rays) overlaps the bulk of the next: throughput mode")
    ap.add_argument("--emulate-parts", type=int, default=0,
                    help="one GPU plays rank --emulate-part of an N-GPU run: renders only that row partition (no gather); "
                         "tools/part_bench.py uses it to project the N-GPU frame time from one GPU")
    ap.add_argument("--emulate-part", type=int, default=0)
    ap.add_argument("--balance", choices=["cyclic", "cost", "auto"], default="auto",
                    help="row blocks -> ranks: block-cyclic, or cost-weighted from one untimed frame's step counts "
                         "(sharding.balance_blocks: the rank that owns the longest ray gets less of the bulk; DESIGN.md 5.1: "
                         "3.82 against 4.03 ms per rank of 8, emulated on one GPU). auto (default) IS block-cyclic: the "
                         "cost-weighted table has never run over RCCL on real devices, so it is opt-in until a hardware run "
                         "confirms the gain; the JSON line records which partition the timed region ran")
    ap.add_argument("--owner-file", default=None, help="(.npy, uint16) a row-block owner table to use as is (emulated ranks)")
    ap.add_argument("--chain-cost", type=float, default=137500.0,
                    help="--balance cost: lane-steps of bulk that take as long as one step of a lone ray (0.55 us / 4.0 ps)")
    ap.add_argument("--pipelined-extra", type=int, default=3,
                    help="after the timed region, time the same K frames again with this many frames in flight and report it "
                         "as `pipelined` (0 / 1: skip)")
    ap.add_argument("--bg-sampling", choices=["lds", "global"], default="global", help="epilogue background path (LT_BG_*)")
    ap.add_argument("--backend", default=None, help="process-group backend (default nccl = RCCL; tests use gloo)")
    ap.add_argument("--stub-render", action="store_true",
                    help="CPU rehearsal of the launcher / gather path: no GPU, rows filled by a formula (tests only)")
    return ap.parse_args(argv)


# ------------------------------------------------------------------------------------------------
# launcher
# ------------------------------------------------------------------------------------------------
def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_ranks(args, argv):
    """Parent of an N-rank run: no GPU call is made in this process (device_count() does not initialise one)."""
    if not args.stub_render and not os.environ.get("LT_BENCH_DEVICES"):
        import torch
        have = torch.cuda.device_count()
        if have < args.gpus:
            print(f"bench.py: --gpus {args.gpus} but this node has {have} GPU(s); refusing to report a "
                  f"{args.gpus}-GPU number from fewer devices", file=sys.stderr)
            return 2
    port = _free_port()
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), LT_BENCH_CHILD="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env))
    worst = 0
    deadline = time.time() + float(os.environ.get("LT_BENCH_TIMEOUT", "1500"))
    for p in procs:
        try:
            rc = p.wait(timeout=max(1.0, deadline - time.time()))
        except subprocess.TimeoutExpired:
            rc = 124
        if rc != 0:
            worst = worst or rc
            for q in procs:               # one rank failed: do not leave the others waiting in a collective
                if q.poll() is None:
                    q.kill()
    return worst


# ------------------------------------------------------------------------------------------------
# untimed extras
# ------------------------------------------------------------------------------------------------
def usable_cpus():
    """CPUs this process may actually use: the affinity mask, cut to the cgroup's CPU quota if it has one (a GPU box
    hands each job a share of a large host: 128 hardware threads visible, 16 usable)."""
    n = len(os.sched_getaffinity(0))
    for path, parse in (("/sys/fs/cgroup/cpu.max", lambda t: t.split()),
                        ("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", lambda t: [t.strip(), open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read().strip()])):
        try:
            q, per = parse(open(path).read())
            if q != "max" and int(q) > 0:
                n = min(n, max(1, int(int(q) / int(per))))
            break
        except (OSError, ValueError):
            continue
    return n


def cpu_baseline(args, fov, integrator):
    """Oracle (CPU port of the reference's tracer, float64, OpenMP), perf build, on a strided subsample of the
    benchmark frame: pixel (k*i, k*j) of the size^2 frame is pixel (i, j) of the (size/k)^2 frame of the same
    camera.  One thread per usable CPU.  `integrator`: "rk4" (metrics.py:570-658) or "dp45" (metrics.py:419-567,
    the reference's production path)."""
    cores = usable_cpus()
    from oracle import oracle
    oracle.set_num_threads(cores, perf_build=True)         # (torch initialised libgomp long ago: the environment is not read again)
    threads = oracle.lib_perf().lto_num_threads()
    kind = args.metric
    kw = dict(integrator="rk4" if integrator == "rk4" else "dp45", perf_build=True)
    oracle.lookup(kind, 1.0, args.a, args.r_obs, 64, 64, fov, fov, **kw)       # compiles; spins the threads up
    t0 = time.perf_counter()
    oracle.lookup(kind, 1.0, args.a, args.r_obs, 256, 256, fov, fov, **kw)
    rate = 256 * 256 / (time.perf_counter() - t0)          # calibration
    stride = 2
    while stride < args.size // 256 and (args.size // stride) ** 2 / rate > args.cpu_seconds:
        stride *= 2
    n = args.size // stride
    t0 = time.perf_counter()
    r = oracle.lookup(kind, 1.0, args.a, args.r_obs, n, n, fov, fov, **kw)
    dt = time.perf_counter() - t0
    return {"value": round(n * n / dt / 1e6, 4), "unit": "Mrays/s", "cores": threads,
            "host_cpus_visible": os.cpu_count(),
            "kind": "port",
            "build": "oracle/lt_oracle.c, gcc " + " ".join(oracle.PERF_FLAGS[:3]) + ", built on this host (perf build)",
            "sample": f"{n}x{n} rays = every {stride}th pixel (x and y) of the {args.size}x{args.size} frame, "
                      f"oracle {kw['integrator']} float64 + OpenMP on {threads} threads (the job's CPU quota), {dt:.1f} s",
            "mean_rhs_evals_per_ray": round(float(r["evals"].mean()), 1)}


def end_to_end(ltrace, cam, met, args, np):
    """What `python image_lens.py` pays per frame: the host-pointer lt_render with an RGBA8 destination only,
    device buffers persistent, destination pinned (ltrace.render's own arrays) or pageable (any numpy array)."""
    o = ltrace.default_opts(integrator=args.integrator, precision=args.precision, schedule=args.schedule)
    res = {}
    import ctypes as C
    for name in ("pinned_dst_ms", "pageable_dst_ms"):
        if name == "pinned_dst_ms":
            rgba = ltrace.pinned_empty((cam.height, cam.width, 4), np.uint8, strict=True)
        else:
            rgba = np.empty((cam.height, cam.width, 4), dtype=np.uint8)
            rgba[:] = 0                                      # touch the pages: a first-touch fault is not PCIe
        st = ltrace.Stats()
        call = lambda: ltrace._check(ltrace.load().lt_render(C.byref(cam), C.byref(met), C.byref(o), None, 3, None, None,
                                                             None, None, None, C.c_void_p(rgba.ctypes.data), C.byref(st)))
        call()
        t0 = time.perf_counter()
        for _ in range(5):
            call()
        res[name] = round((time.perf_counter() - t0) / 5 * 1e3, 3)
    res["what"] = "host-pointer lt_render, RGBA8 out only, mean of 5 frames (device time + PCIe + host copy)"
    return res


def longest_chain(ltrace, cam, met, args, steps_host, np):
    """The serial chain that bounds small frames and strong scaling: the frame's longest ray traced ALONE on the chip."""
    size = cam.width
    iy, ix = divmod(int(np.argmax(steps_host)), size)
    n_steps = int(steps_host[iy, ix])
    f = (size / 2.0) / np.tan(cam.hfov / 2)
    x, y = (ix - size / 2.0) / f, (iy - size / 2.0) / f      # image_lens.py:141-142, psi = (0, 0)
    den = np.sqrt(1.0 + x * x + y * y)
    alpha = float(np.float32(np.arccos(1.0 / den)))
    theta = float(np.arctan2(x / den, y / den))
    x_lo, x_hi = abs((0 - size / 2.0) / f), abs((size - 1 - size / 2.0) / f)
    refine = int(abs(x) <= 0.07 * max(x_lo, x_hi))            # image_lens.py:210-214

    def run(al, th, rf):
        fa, w = np.empty(64), np.empty(64, dtype=np.int64)
        ev = np.empty(64, dtype=np.uint32)
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter()
            ltrace.trace_batch_kerr(1.0, args.a, args.r_obs, np.full(64, al), np.full(64, th), np.pi / 2,
                                    max(5000.0, 6.0 * args.r_obs), np.full(64, rf, dtype=np.uint8), fa, w,
                                    integrator=args.integrator, precision=args.precision, out_rhs_evals=ev)
            best = min(best, time.perf_counter() - t0)
        return best, int(ev[0])
    base, _ = run(0.3, 1.0, 0)                                # a short ray: the call's fixed cost
    t, _ = run(alpha, theta, refine)
    alone_ms = (t - base) * 1e3
    return {"longest_ray_steps": n_steps, "pixel": [iy, ix], "alone_ms": round(alone_ms, 3),
            "us_per_step": round(alone_ms * 1e3 / max(n_steps, 1), 4),
            "what": "the frame's longest ray traced alone on the chip (one wavefront): no launch of this frame, "
                    "on any number of GPUs, ends before it does"}


def load_profile_json(name):
    try:
        with open(os.path.join(ROOT, "profiles", name)) as f:
            return json.load(f)
    except (OSError, ValueError):
        return {}


# ------------------------------------------------------------------------------------------------
# one rank
# ------------------------------------------------------------------------------------------------
def stub_main(args, world, rank):
    """Launcher / gather rehearsal on CPU (tests): same process-group setup, FrameGather, timing and JSON
    plumbing as the real path; rows are filled by a formula instead of a render."""
    import torch
    import torch.distributed as dist
    import sharding
    if world > 1:
        dist.init_process_group(backend=args.backend or "gloo")
        if dist.get_world_size() != args.gpus:
            raise SystemExit(f"process group has {dist.get_world_size()} ranks, --gpus {args.gpus}")
    size = args.size
    fg = sharding.FrameGather(size, size, 4, torch.uint8, "cpu", args.row_block, world, rank)
    rows = sharding.global_row_index(size, args.row_block, world, rank).to(torch.int64)
    val = ((rows[:, None, None] * 131 + torch.arange(size)[None, :, None] * 7 + torch.arange(4)[None, None, :] * 3) % 251)
    ok = True
    t0 = time.perf_counter()
    for _ in range(args.warmup + args.steps):
        fg.local_view().copy_(val.to(torch.uint8))
        full = fg.gather()
    elapsed = time.perf_counter() - t0
    if rank == 0:
        r = torch.arange(size)
        expect = ((r[:, None, None] * 131 + torch.arange(size)[None, :, None] * 7 + torch.arange(4)[None, None, :] * 3) % 251)
        ok = bool(torch.equal(full, expect.to(torch.uint8)))
        print(json.dumps({"metric": "Mrays/s", "value": 0.0, "unit": "Mrays/s", "n_gpus": world, "stub": True,
                          "rccl_world": dist.get_world_size() if world > 1 else 1, "frame_ok": ok,
                          "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed * 1e3 / max(args.steps, 1), 3)}),
              flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return 0 if ok else 1


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    args = parse(argv)
    env_world = os.environ.get("WORLD_SIZE")
    if env_world is None and args.gpus > 1:
        return launch_ranks(args, argv)
    world = int(env_world or "1")
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: start it with --nproc-per-node {args.gpus}, "
              f"or plainly and let it start its own ranks", file=sys.stderr)
        return 2
    if args.stub_render:
        return stub_main(args, world, rank)

    import numpy as np
    import torch
    import torch.distributed as dist
    import ltrace
    import sharding

    if not torch.cuda.is_available():
        print("bench.py needs a GPU (no CPU fallback)", file=sys.stderr)
        return 2
    # LT_BENCH_DEVICES="0,0": rehearsal of the N-rank path on fewer GPUs (ranks share devices; only with --backend gloo,
    # RCCL refuses two ranks on one device).  Never set by the driver; the JSON says so when it is.
    dev_map = os.environ.get("LT_BENCH_DEVICES")
    dev_index = int(dev_map.split(",")[local_rank]) if dev_map else local_rank
    if dev_index >= torch.cuda.device_count():
        print(f"bench.py: rank {rank} wants GPU {dev_index} but the node has {torch.cuda.device_count()}", file=sys.stderr)
        return 2
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    rccl_world = 1
    staged = False   # gloo moves tensors through the host: all_gather / gather of device tensors are staged by hand
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = args.backend or "nccl"
        if dev_map and backend == "nccl":
            print("bench.py: LT_BENCH_DEVICES is a rehearsal switch and needs --backend gloo", file=sys.stderr)
            return 2
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=dev)
        else:
            dist.init_process_group(backend=backend)
            staged = True
        rccl_world = dist.get_world_size()
        if rccl_world != args.gpus:
            print(f"bench.py: the process group has {rccl_world} ranks, --gpus {args.gpus}", file=sys.stderr)
            return 2
    # who this rank is: one line per rank on stderr, and (N > 1) every rank's device in the JSON line, so that a
    # multi-GPU number can be checked against the devices it ran on
    ident = {"rank": rank, "device_index": dev_index}
    try:
        props = torch.cuda.get_device_properties(dev)
        ident["name"] = props.name
        ident["uuid"] = str(getattr(props, "uuid", ""))
        ident["rccl"] = ".".join(str(x) for x in torch.cuda.nccl.version()) if world > 1 and not staged else None
    except Exception as e:                       # identification must not cost the run
        ident["error"] = f"{type(e).__name__}: {e}"
    print(f"bench.py: rank {rank}/{world} pid {os.getpid()} {ident}", file=sys.stderr, flush=True)
    idents = [ident]
    if world > 1:
        idents = [None] * world
        dist.all_gather_object(idents, ident)
        if not dev_map:
            # two ranks on one device index cannot be a valid run; equal UUIDs alone only earn a warning (a driver that
            # reports no per-device UUID must not cost the multi-GPU number) -- the JSON carries what every rank saw
            if len({(i or {}).get("device_index") for i in idents}) != world:
                print("bench.py: two ranks report the same device index", file=sys.stderr)
                return 2
            uu = [(i or {}).get("uuid") for i in idents]
            if all(uu) and len(set(uu)) != world:
                print(f"bench.py: warning: ranks report equal device UUIDs {uu}", file=sys.stderr)

    def all_reduce(t, op):
        if not staged:
            dist.all_reduce(t, op=op)
        else:
            c = t.cpu(); dist.all_reduce(c, op=op); t.copy_(c)

    def all_gather(out, t):
        if not staged:
            dist.all_gather(out, t)
        else:
            co = [torch.zeros_like(t, device="cpu") for _ in out]
            dist.all_gather(co, t.cpu())
            for o_, c_ in zip(out, co):
                o_.copy_(c_)

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    size = args.size
    fov = float(np.radians(40.0))
    cam = ltrace.Camera(size, size, fov, fov, 0.0, 0.0, args.r_obs, np.pi / 2)
    met = ltrace.Metric(ltrace.METRIC_KERR if args.metric == "kerr" else ltrace.METRIC_SCHWARZSCHILD, 0, 1.0,
                        args.a if args.metric == "kerr" else 0.0)
    rb = args.row_block
    n_parts, part = (args.emulate_parts, args.emulate_part) if (args.emulate_parts and world == 1) else (world, rank)
    spin = args.a if args.metric == "kerr" else 0.0
    bid = ltrace.build_id()
    main_stream = torch.cuda.current_stream(dev)

    # ---- row blocks -> ranks.  `auto` IS block-cyclic: the cost-weighted table (3.82 against 4.03 ms for the
    # slowest rank of 8, emulated on one GPU) has never run over RCCL on real devices, so it stays opt-in (--balance cost)
    # until a hardware run confirms it; and when asked for, any failure on any rank sends ALL ranks back to block-cyclic.
    owner, balance_used = None, "cyclic"
    if args.owner_file:
        owner = np.load(args.owner_file).astype(np.uint16)
        balance_used = "owner-file"
    elif args.balance == "cost" and world > 1 and args.metric == "kerr":     # (an emulated rank takes --owner-file)
        nb = (size + rb - 1) // rb
        cost_t = torch.zeros(nb, dtype=torch.int64, device=dev)
        chain_t = torch.zeros(nb, dtype=torch.int64, device=dev)
        ok = 1
        try:
            o0 = ltrace.default_opts(integrator=args.integrator, precision=args.precision, schedule=args.schedule,
                                     row_block=rb, n_parts=n_parts, part=part)
            o0.stream = main_stream.cuda_stream
            r0 = ltrace.local_rows(size, rb, n_parts, part)
            d_steps0 = torch.empty((r0, size), dtype=torch.int32, device=dev)
            ltrace.render_dev(cam, met, o0, d_steps=d_steps0.data_ptr())
            torch.cuda.synchronize(dev)
            st = d_steps0.to(torch.int64)
            mine_blocks = torch.from_numpy(ltrace.global_rows(size, rb, n_parts, part)[::rb] // rb).to(dev)
            per_row_sum, per_row_max = st.sum(dim=1), st.max(dim=1).values
            pad = (-r0) % rb
            if pad:
                per_row_sum = torch.cat([per_row_sum, per_row_sum.new_zeros(pad)])
                per_row_max = torch.cat([per_row_max, per_row_max.new_zeros(pad)])
            cost_t[mine_blocks] = per_row_sum.view(-1, rb).sum(dim=1)
            chain_t[mine_blocks] = per_row_max.view(-1, rb).max(dim=1).values
            del d_steps0, st
        except Exception as e:
            print(f"bench.py: rank {rank}: cost pre-pass failed ({type(e).__name__}: {e}); block-cyclic partition", file=sys.stderr)
            ok = 0
        okt = torch.tensor([ok], dtype=torch.int64, device=dev)
        all_reduce(okt, dist.ReduceOp.MIN)
        all_reduce(cost_t, dist.ReduceOp.SUM)
        all_reduce(chain_t, dist.ReduceOp.SUM)
        if int(okt.item()) == 1:
            owner = sharding.balance_blocks(cost_t.cpu().numpy(), chain_t.cpu().numpy(), n_parts, chain_cost=args.chain_cost)
            if len(owner) != nb or int(owner.max()) >= n_parts:      # (a table every rank computes alike, or none)
                owner = None
        if owner is not None:
            # one untimed frame through the table-mode render on every rank before it is trusted (the gather that follows
            # is the same collective either way); a rank that fails takes every rank back to block-cyclic
            ok = 1
            try:
                o1 = ltrace.default_opts(integrator=args.integrator, precision=args.precision, schedule=args.schedule,
                                         row_block=rb, n_parts=n_parts, part=part, block_owner=owner)
                o1.stream = main_stream.cuda_stream
                r1 = len(ltrace.owned_rows(size, rb, owner, part))
                probe = torch.empty((max(r1, 1), size, 4), dtype=torch.uint8, device=dev)
                ltrace.render_dev(cam, met, o1, d_rgba=probe.data_ptr())
                torch.cuda.synchronize(dev)
                del probe
            except Exception as e:
                print(f"bench.py: rank {rank}: table-mode render failed ({type(e).__name__}: {e}); block-cyclic partition", file=sys.stderr)
                ok = 0
            okt = torch.tensor([ok], dtype=torch.int64, device=dev)
            all_reduce(okt, dist.ReduceOp.MIN)
            if int(okt.item()) != 1:
                owner = None
        balance_used = "cost" if owner is not None else "cyclic (cost-weighted table unavailable: fell back)"
    if owner is not None:
        rows_max = max(len(ltrace.owned_rows(size, rb, owner, p)) for p in range(n_parts))
    else:
        rows_max = max(ltrace.local_rows(size, rb, n_parts, p) for p in range(n_parts))

    d_stats = torch.zeros(ltrace.STAT_WORDS, dtype=torch.int64, device=dev)
    d_bg = None
    if args.background:
        g = torch.Generator(device="cpu").manual_seed(0)
        d_bg = (torch.randint(0, 256, (size, size, 3), generator=g, dtype=torch.uint8).to(torch.float32) / 255.0).to(dev)
    torch.cuda.synchronize(dev)

    def workload_name(integ):
        w = f"{args.metric}_a{spin}_shadow_{size}x{size}_r{args.r_obs:g}_{integ}"
        return w + ("_lensed_background" if args.background else "")

    # --------------------------------------------------------------------------------------------
    # one timed region: W warm-up frames, then K frames between barrier + synchronize on both sides
    # --------------------------------------------------------------------------------------------
    def run_region(integ, prec, K, Wu, F, pipelined_extra):
        def make_sets(nf):
            out = []
            for f in range(nf):
                st = main_stream if nf == 1 else torch.cuda.Stream(dev)
                o = ltrace.default_opts(integrator=integ, precision=prec, schedule=args.schedule,
                                        row_block=rb, n_parts=n_parts, part=part, timing=1, block_owner=owner,
                                        bg_sampling=ltrace.BG_LDS_TILES if args.bg_sampling == "lds" else ltrace.BG_GLOBAL)
                o.stream = st.cuda_stream
                fg = (sharding.FrameGather(size, size, 4, torch.uint8, dev, rb, world, rank, owner=owner) if n_parts == world else
                      sharding.FrameGather(rows_max, size, 4, torch.uint8, dev, rows_max, 1, 0))          # emulated rank: no gather
                out.append(dict(stream=st, opts=o, fg=fg,   # RGBA8 framebuffer
                                fa=torch.empty((rows_max, size), dtype=torch.float32, device=dev),
                                w=torch.empty((rows_max, size), dtype=torch.int16, device=dev)))
            return out

        def drop_sets(ss):
            # a stream made for a region owns ray records inside the library (0.8 GB at 4096^2 float32): give them back
            torch.cuda.synchronize(dev)
            for b in ss:
                if b["stream"] is not main_stream:
                    ltrace.release_stream(b["stream"].cuda_stream)

        sets = make_sets(F)
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(3 * max(K, 1))]   # frame start / rendered / gathered

        def step(i, k, ss):
            b = ss[k % len(ss)]
            st = b["stream"]
            with torch.cuda.stream(st):
                if i is not None:
                    ev[3 * i].record(st)
                ltrace.render_dev(cam, met, b["opts"], d_bg=d_bg.data_ptr() if d_bg is not None else 0, bg_channels=3,
                                  d_fa=b["fa"].data_ptr(), d_w=b["w"].data_ptr(), d_rgba=b["fg"].local_view().data_ptr(),
                                  d_stats=d_stats.data_ptr())
                if i is not None:
                    ev[3 * i + 1].record(st)
                full = b["fg"].gather(st.cuda_stream)   # N > 1: RCCL exchange to rank 0 + row un-permute there
                if i is not None:
                    ev[3 * i + 2].record(st)
            return full

        for j in range(Wu):
            step(None, j, sets)
        fence()
        ltrace.timing_collect()            # drop warm-up events
        d_stats.zero_()
        fence()
        t0 = time.perf_counter()
        for i in range(K):
            step(i, i, sets)
        fence()
        elapsed = time.perf_counter() - t0
        tm = ltrace.timing_collect()

        steps = max(K, 1)
        render_ms = sum(ev[3 * i].elapsed_time(ev[3 * i + 1]) for i in range(K)) / steps
        gather_ms = sum(ev[3 * i + 1].elapsed_time(ev[3 * i + 2]) for i in range(K)) / steps
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        counters = d_stats.clone()
        mine = torch.tensor([tm["prologue_ms"] / steps, tm["integrate_ms"] / steps, tm["epilogue_ms"] / steps,
                             render_ms, gather_ms], dtype=torch.float64, device=dev)
        per_rank = [mine]
        if world > 1:
            all_reduce(t, dist.ReduceOp.MAX)
            per_rank = [torch.zeros_like(mine) for _ in range(world)]
            all_gather(per_rank, mine)
        elapsed = float(t.item())
        my = [int(x) for x in counters.tolist()]
        if world > 1:
            all_reduce(counters, dist.ReduceOp.SUM)
        c = [int(x) for x in counters.tolist()]
        per_rank = [[float(x) for x in r.tolist()] for r in per_rank]
        # counters of every rank's integrate kernel (the slowest rank's launch is the one priced)
        mine_k = torch.tensor([my[ltrace.STAT_WAVE_ITERS], my[ltrace.STAT_WAVES], my[ltrace.STAT_CLK_CYCLES],
                               my[ltrace.STAT_CLK_TICKS], my[ltrace.STAT_STEPS], my[ltrace.STAT_RAYS]], dtype=torch.int64, device=dev)
        all_k = [mine_k]
        if world > 1:
            all_k = [torch.zeros_like(mine_k) for _ in range(world)]
            all_gather(all_k, mine_k)
        all_k = [[int(x) for x in k.tolist()] for k in all_k]
        rays_per_frame = c[ltrace.STAT_RAYS] // steps
        R = dict(integrator=integ, precision=prec, steps=K, warmup=Wu, frames_in_flight=F, elapsed=elapsed,
                 ms_per_step=elapsed / steps * 1e3, rays_per_frame=rays_per_frame,
                 value=rays_per_frame / (elapsed / steps) / 1e6, counters=c, per_rank=per_rank, all_k=all_k,
                 slow=max(range(world), key=lambda r: per_rank[r][1]), pipelined=None)

        # second region, same K and W, frames pipelined (every rank takes part: the exchange is collective)
        if F == 1 and pipelined_extra > 1:
            psets = None
            try:
                psets = make_sets(pipelined_extra)
                for j in range(max(Wu, pipelined_extra)):
                    step(None, j, psets)
                fence()
                t0 = time.perf_counter()
                for i in range(K):
                    step(None, i, psets)
                fence()
                tp = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev)
                if world > 1:
                    all_reduce(tp, dist.ReduceOp.MAX)
                ltrace.timing_collect()
                pel = float(tp.item())
                R["pipelined"] = {"frames_in_flight": pipelined_extra, "value": round(rays_per_frame / (pel / steps) / 1e6, 2), "unit": "Mrays/s",
                                  "ms_per_step": round(pel / steps * 1e3, 4),
                                  "what": f"the same {K} frames after the same warm-up, frame i on stream i % {pipelined_extra} with its own buffers: "
                                          "the tail of a frame (a few lone wavefronts finishing its longest rays) overlaps the bulk of the next. "
                                          "Throughput of a frame SEQUENCE; `value` above is frames strictly one after the other"}
            except Exception as e:      # the headline line must not depend on the extra region
                R["pipelined"] = {"error": f"{type(e).__name__}: {e}"}
            if psets:
                drop_sets(psets)
        drop_sets(sets)
        return R

    # --------------------------------------------------------------------------------------------
    # the integrate kernel of a region against its VALU-issue roofline (executed work, PMC-counted)
    # --------------------------------------------------------------------------------------------
    def roofline_of(R):
        integ, prec, steps = R["integrator"], R["precision"], max(R["steps"], 1)
        workload = workload_name(integ)
        kernel = (f"k_kerr_{args.schedule}<{integ}>" if args.metric == "kerr" else "k_schw_rk4_direct")
        key = f"{workload}|f{prec}|{args.schedule}|parts{n_parts}"
        if n_parts != world:
            key += f"|part{part}"
        slow = R["slow"]
        k_iters, k_waves, k_cyc, k_ticks, k_steps, k_rays = R["all_k"][slow]
        integ_ms = R["per_rank"][slow][1]
        iters_per_launch = k_iters / steps
        vc = load_profile_json("valu_counts.json")
        wl = vc.get("workloads") or {}
        rec = wl.get(key) or wl.get(f"{workload}|f{prec}|{args.schedule}|parts1")
        valu_src, valu_per_iter = None, None
        if rec and rec.get("wave_iters"):
            valu_per_iter = rec["valu_insts"] / rec["wave_iters"]
            rec_bid = rec.get("build_id") or vc.get("build_id")
            valu_src = (f"{rec.get('source', 'profiles/valu_counts.json')}: SQ_INSTS_VALU {rec['valu_insts']} / wave_iters "
                        f"{rec['wave_iters']} per launch" + ("" if rec_bid == bid else f"; STALE: measured on build {rec_bid}, this is {bid}"))
        peak = PEAK_FP32_VALU_TFLOPS if prec == 32 else PEAK_FP64_VALU_TFLOPS
        cyc = CYCLES_PER_VALU[prec]
        clock_mhz = k_cyc / k_ticks * 100.0 if k_ticks else None
        roof = {"bound": "valu_issue_fp32" if prec == 32 else "valu_issue_fp64", "kernel": kernel,
                "achieved": None, "peak": peak, "unit": "TFLOP/s", "frac": None, "frac_at_held_clock": None, "traffic": None,
                "avg_launch_ms": round(integ_ms, 4), "priced_rank": slow}
        if valu_per_iter and integ_ms > 0:
            valu = valu_per_iter * iters_per_launch               # wave-instructions this launch issued
            slots_per_s = valu / (integ_ms * 1e-3)
            peak_slots = N_SIMD * NOMINAL_CLOCK_HZ / cyc
            lane_flops = 64 * 2                                   # one FMA per lane per issue slot
            roof["achieved"] = round(slots_per_s * lane_flops / 1e12, 2)   # FMA-equivalent: every issue slot priced as one FMA
            roof["frac"] = round(slots_per_s / peak_slots, 4)
            roof["frac_at_held_clock"] = round(slots_per_s / (N_SIMD * clock_mhz * 1e6 / cyc), 4) if clock_mhz else None
            # the same over the whole timed region (prologue, epilogue, gaps and -- with frames in flight -- overlap
            # included): wave-instructions the integrate kernels of this rank issued / (wall time x issue peak)
            roof["region_issue_frac"] = round(valu * steps / R["elapsed"] / peak_slots, 4)
            if R["frames_in_flight"] > 1:
                roof["note"] = (f"{R['frames_in_flight']} frames in flight: launches of consecutive frames overlap, so avg_launch_ms (HIP events around "
                                "each launch) includes time shared with the neighbouring frame and `frac` understates; "
                                "`region_issue_frac` is the chip-level figure for this mode")
            roof["executed"] = {"valu_wave_insts_per_launch": int(valu), "wave_iters_per_launch": int(iters_per_launch),
                                "valu_per_wave_iter": round(valu_per_iter, 2), "waves": int(k_waves / steps),
                                "issue_cycles_per_inst": cyc, "source": valu_src,
                                "clock_mhz_held": round(clock_mhz, 1) if clock_mhz else None,
                                "frac_at_held_clock": roof["frac_at_held_clock"]}
        else:
            roof["executed"] = {"wave_iters_per_launch": int(iters_per_launch), "source": "no VALU count for this workload under profiles/ "
                                "(tools/pmc_counts.py); frac left null rather than guessed"}
        # --- the reference's as-written count, for the record
        slow_steps, slow_rays = k_steps / steps, k_rays / steps
        if args.metric == "kerr":
            flops = (slow_steps * F_DP45_ATTEMPT + slow_rays * (F_KERR_FIXED + 188)) if integ != "rk4" else \
                    (slow_steps * F_RK4_STEP + slow_rays * F_KERR_FIXED)
        else:
            flops = slow_steps * F_SCHW_STEP + slow_rays * F_SCHW_FIXED
        aw = flops / (integ_ms * 1e-3) / 1e12 if integ_ms > 0 else 0.0
        roof["algorithmic_as_written"] = {"flops_per_launch": int(flops), "tflops": round(aw, 2), "of_peak": round(aw / peak, 4),
                                          "note": "the reference's un-simplified op count (832 per RK4 step, 1448 per DP45 attempt); the kernel "
                                                  "executes a fraction of it, so this can exceed 1 and is not a roofline fraction"}
        # --- HBM traffic of the integrate kernel: only from a PMC pass of THIS build on THIS workload
        tr = load_profile_json("hbm_traffic.json")
        trec = (tr.get("workloads") or {}).get(key)
        if trec and (trec.get("build_id") or tr.get("build_id")) == bid:
            roof["traffic"] = trec["bytes_per_launch"]
            roof["traffic_source"] = trec.get("source")
        # 16 B read + 32 B written per ray record (three 4-vectors of the integrator's type), padded tiles included
        # (direct schedule: waves x 64 records; the queue schedule's few persistent waves sweep the same records)
        roof["algorithmic_bytes_per_launch"] = int(max(k_waves / steps * 64, slow_rays) * 12 * (4 if prec == 32 else 8))
        roof["other_kernels_ms"] = {"prologue": round(R["per_rank"][slow][0], 4), "epilogue": round(R["per_rank"][slow][2], 4)}
        return roof, workload, key

    # what ONE rank of an n-GPU run renders, each rank in turn on this one GPU under benchmark conditions
    def projected_ranks(integ, prec, frames=4):
        res = {}
        for n in (2, 4, 8):
            per = []
            for p in range(n):
                rows = ltrace.local_rows(size, rb, n, p)
                rgba = torch.empty((max(rows, 1), size, 4), dtype=torch.uint8, device=dev)
                o = ltrace.default_opts(integrator=integ, precision=prec, schedule=args.schedule, row_block=rb, n_parts=n, part=p)
                o.stream = main_stream.cuda_stream
                for _ in range(2):
                    ltrace.render_dev(cam, met, o, d_rgba=rgba.data_ptr())
                torch.cuda.synchronize(dev)
                t0 = time.perf_counter()
                for _ in range(frames):
                    ltrace.render_dev(cam, met, o, d_rgba=rgba.data_ptr())
                torch.cuda.synchronize(dev)
                per.append((time.perf_counter() - t0) / frames * 1e3)
                del rgba
            worst = max(per)
            res[str(n)] = {"frame_ms_per_rank": [round(x, 3) for x in per], "slowest_rank_ms": round(worst, 3),
                           "mrays_per_s_before_gather": round(size * size / worst / 1e3, 1)}
        res["what"] = (f"PROJECTION from one GPU, not a multi-GPU measurement: rank p of n (block-cyclic {rb}-row blocks) renders its rows "
                       f"alone on this GPU, {frames} frames back to back after 2 warm-up frames; an n-GPU frame takes at least the slowest "
                       "rank's time plus the gather (4 MiB x rows / 1024 per peer over one xGMI link each)")
        return res

    F = max(1, args.frames_in_flight)
    extras = set() if args.no_extras else {x.strip() for x in args.extras.split(",") if x.strip()}
    H = run_region(args.integrator, args.precision, args.steps, args.warmup, F, args.pipelined_extra if "pipelined" in extras else 0)

    # ---- the reference's production path (metrics.py:419-567 via :1128-1132: DP45, float64), same frame, same K and W.
    # It is what `python image_lens.py --a 0.9` integrates with and what the plugin surface defaults to (metrics.py of
    # the product: dp45_exact); the headline above is the fixed-step float32 integrator BASELINE.json's north star names.
    P = None
    want_prod = ("production" in extras and not args.no_production_path and args.metric == "kerr" and args.integrator == "rk4"
                 and not args.background and n_parts == world)
    if want_prod:
        P = run_region("dp45_exact", 64, args.steps, min(args.warmup, 3), 1, 0)

    # untimed extras (rank 0, one GPU): longest-ray chain, host-pointer end-to-end frame, per-rank projection
    chain = e2e = proj = None
    if rank == 0 and world == 1 and n_parts == 1 and args.metric == "kerr":
        if "chain" in extras:
            d_steps = torch.empty((size, size), dtype=torch.int32, device=dev)
            o1 = ltrace.default_opts(integrator=args.integrator, precision=args.precision, schedule=args.schedule)
            o1.stream = main_stream.cuda_stream
            ltrace.render_dev(cam, met, o1, d_steps=d_steps.data_ptr())
            torch.cuda.synchronize(dev)
            chain = longest_chain(ltrace, cam, met, args, d_steps.cpu().numpy(), np)
            del d_steps
        if not args.background and "e2e" in extras:
            e2e = end_to_end(ltrace, cam, met, args, np)
        if not args.background and "projection" in extras:
            try:
                proj = {args.integrator: projected_ranks(args.integrator, args.precision)}
                if P is not None:
                    proj["dp45_exact"] = projected_ranks("dp45_exact", 64)
            except Exception as e:
                proj = {"error": f"{type(e).__name__}: {e}"}

    if rank == 0:
        steps = max(args.steps, 1)
        roof, workload, key = roofline_of(H)
        c = H["counters"]
        out = {
            "metric": "Mrays/s", "value": round(H["value"], 2), "unit": "Mrays/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(H["ms_per_step"], 4),
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f32" if args.precision == 32 else "f64", "data": "synthetic",
            "config": {"workload": workload,
                       "rays_per_frame": H["rays_per_frame"], "schedule": args.schedule, "build_id": bid, "profile_key": key,
                       "row_partition": (f"block-cyclic {rb} rows x {n_parts}" if owner is None else
                                         f"cost-weighted {rb}-row blocks x {n_parts} (rows per rank "
                                         f"{[len(ltrace.owned_rows(size, rb, owner, p)) for p in range(n_parts)]})") + (f" (emulated rank {part} on one GPU)" if n_parts != world else ""),
                       "balance": {"requested": args.balance, "used": balance_used},
                       "gather": ("rccl send/recv to rank 0, exact partition sizes" if not staged else "gloo, staged through the host (rehearsal)") if world > 1 else "none",
                       **({"rehearsal_devices": dev_map} if dev_map else {}),
                       "frames_in_flight": F,
                       "mean_rk4_steps_per_ray": round(c[ltrace.STAT_STEPS] / steps / max(H["rays_per_frame"], 1), 2),
                       "escaped": c[ltrace.STAT_ESCAPED] // steps, "captured": c[ltrace.STAT_CAPTURED] // steps,
                       "invalid": c[ltrace.STAT_INVALID] // steps,
                       **({"bg_sampling": args.bg_sampling, "bg_groups_lds": c[ltrace.STAT_BG_TILES_LDS] // steps,
                           "bg_groups_global": c[ltrace.STAT_BG_TILES_GLOBAL] // steps} if args.background else {})},
            "roofline": roof,
            "ranks": {"rccl_world": rccl_world,
                      "integrate_ms": [round(r[1], 3) for r in H["per_rank"]],
                      "render_ms": [round(r[3], 3) for r in H["per_rank"]],
                      "gather_ms": [round(r[4], 3) for r in H["per_rank"]],
                      "devices": idents},
        }
        if H["pipelined"]:
            out["pipelined"] = H["pipelined"]
        if chain:
            out["chain_floor"] = chain
        if e2e:
            out["end_to_end_ms"] = e2e
        if proj:
            out["projected_ranks"] = proj
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(args, fov, args.integrator)
        if P is not None:
            proof, pworkload, pkey = roofline_of(P)
            pc = P["counters"]
            psteps = max(P["steps"], 1)
            prod = {"what": "the same frame with the reference's production integrator (DP45 rtol 1e-8 / atol 1e-10, float64, "
                            "metrics.py:419-567; step controller evaluated in float64 as written there) -- the plugin default",
                    "metric": "Mrays/s", "value": round(P["value"], 2), "unit": "Mrays/s", "dtype": "f64",
                    "steps": P["steps"], "warmup": P["warmup"], "ms_per_step": round(P["ms_per_step"], 4),
                    "config": {"workload": pworkload, "profile_key": pkey, "rays_per_frame": P["rays_per_frame"],
                               "mean_dp45_attempts_per_ray": round(pc[ltrace.STAT_STEPS] / psteps / max(P["rays_per_frame"], 1), 2),
                               "mean_rhs_evals_per_ray": round(pc[ltrace.STAT_RHS_EVALS] / psteps / max(P["rays_per_frame"], 1), 1),
                               "escaped": pc[ltrace.STAT_ESCAPED] // psteps, "captured": pc[ltrace.STAT_CAPTURED] // psteps,
                               "invalid": pc[ltrace.STAT_INVALID] // psteps},
                    "roofline": proof,
                    "ranks": {"integrate_ms": [round(r[1], 3) for r in P["per_rank"]],
                              "render_ms": [round(r[3], 3) for r in P["per_rank"]],
                              "gather_ms": [round(r[4], 3) for r in P["per_rank"]]}}
            if not args.no_cpu_baseline and world == 1:
                prod["cpu_baseline"] = cpu_baseline(args, fov, "dp45")
            out["production_path"] = prod
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
