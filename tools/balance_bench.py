#!/usr/bin/env python3
"""Cost-weighted row-block assignment (sharding.balance_blocks) against block-cyclic, projected from ONE GPU: one full
frame gives per-block step totals and longest rays; for N = 2, 4, 8 the owner table is computed and every rank's
partition is rendered back to back under benchmark conditions (bench.py --emulate-parts N --emulate-part p
--owner-file ...).  The N-GPU frame time is the slowest rank's.   usage: balance_bench.py [size]"""
import json, os, subprocess, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "light-path-tracer_amd"))
import ltrace
import torch   # noqa: F401  (sharding imports it)
import sharding
size = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
rb = 16
fov = np.radians(40.0)
cam = ltrace.Camera(size, size, fov, fov, 0.0, 0.0, 50.0, np.pi / 2)
out = ltrace.render(cam, ltrace.Metric(1, 0, 1.0, 0.9), ltrace.default_opts(precision=32), want=("steps",))
st = out["steps"].astype(np.int64).reshape(size // rb, rb * size)
cost, chain = st.sum(axis=1), st.max(axis=1)
ltrace.shutdown()


def rank_ms(n, p, owner_file):
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--size", str(size), "--no-cpu-baseline", "--no-extras", "--steps", "10",
           "--emulate-parts", str(n), "--emulate-part", str(p)] + (["--owner-file", owner_file] if owner_file else [])
    d = json.loads(subprocess.run(cmd, capture_output=True, text=True).stdout.strip().splitlines()[-1])
    return d["ms_per_step"], d["config"]["rays_per_frame"]


with tempfile.TemporaryDirectory() as tmp:
    for n in (2, 4, 8):
        owner = sharding.balance_blocks(cost, chain, n, chain_cost=137500.0)
        f = os.path.join(tmp, f"owner{n}.npy")
        np.save(f, owner)
        cyc = [rank_ms(n, p, None)[0] for p in range(n)]
        bal = [rank_ms(n, p, f) for p in range(n)]
        print(f"n_parts={n}: block-cyclic frame ms per rank {[round(x, 2) for x in cyc]} -> slowest {max(cyc):.2f} ms = {size * size / max(cyc) / 1e3:.0f} Mrays/s")
        print(f"            cost-weighted frame ms per rank {[round(x[0], 2) for x in bal]} -> slowest {max(x[0] for x in bal):.2f} ms = "
              f"{size * size / max(x[0] for x in bal) / 1e3:.0f} Mrays/s; rows per rank {[x[1] // size for x in bal]}", flush=True)
