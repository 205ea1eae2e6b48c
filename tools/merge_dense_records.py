#!/usr/bin/env python3
"""profiles/valu_counts.json += the k_dense_tracks records tools/prof_dense.sh wrote (gpurun_out/prof_dense_{binned,caller}/valu_record.json).
Used by tools/refresh_profiles.sh, on the box (so that the dense_bench.py lines that follow are priced with this build's counts) and in `collect`."""
import json
import os

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
p = os.path.join(ROOT, "profiles", "valu_counts.json")
d = json.load(open(p))
for tag in ("binned", "caller"):
    f = os.path.join(ROOT, "gpurun_out", f"prof_dense_{tag}", "valu_record.json")
    if os.path.exists(f):
        d.setdefault("workloads", {}).update(json.load(open(f)))
json.dump(d, open(p, "w"), indent=1)
