#!/usr/bin/env python3
"""VALU issue cost table: true cycles (at the measured in-kernel clock) per wave-instruction per SIMD."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import probes as ltrace   # the diagnostic build (libltrace_probes.so)
n = ltrace.valu_issue_probe_count()
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
print(f"{'instruction':18s} {'data':9s} " + " ".join(f"{'w=' + str(w) + ' cyc':>9s} {'MHz':>6s}" for w in (1, 2, 4, 8)))
first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
last = int(sys.argv[3]) if len(sys.argv) > 3 else n
for i in range(first, last):
    for const in (False,):
        row = []
        for w in (1, 2, 4, 8):
            name, ns, clk = ltrace.valu_issue_probe(i, w, iters, const)
            row.append((ns * clk * 1e-3, clk))
        print(f"{name:18s} {'constant' if const else 'varying':9s} " + " ".join(f"{c:9.2f} {k:6.0f}" for c, k in row))
