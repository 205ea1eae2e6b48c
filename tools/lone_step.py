#!/usr/bin/env python3
"""Cycles one SIMD spends per RK4 wave-step at 1, 2, 4, 8 resident waves per SIMD (lt_rk4_step_probe):
the serial-chain speed of a lone long ray is the 1-wave figure."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import probes as ltrace   # the diagnostic build (libltrace_probes.so)
for w in (1, 2, 4, 8):
    c, mhz = ltrace.rk4_step_probe(32, w, 20000)
    print(f"waves/SIMD {w}: {c:8.1f} cycles per wave-step on the SIMD, {c * w:8.1f} per step of one wave, clock {mhz:.0f} MHz,"
          f" {c * w / mhz:.3f} us per step")
