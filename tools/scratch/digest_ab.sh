#!/bin/bash
# outputs of tests/ghost_lanes_check.py under two builds (and with / without the lone-wave phase): all digests must be equal
L=$PWD/light-path-tracer_amd/lib
for lib in libltrace_prev.so libltrace_hip.so; do
  for d in 2000000000 1; do
    echo "$lib LT_D_LONG=$d: $(LTRACE_LIB=$L/$lib LT_D_LONG=$d LT_Q_LONG=2000000000 python tests/ghost_lanes_check.py | grep digest)"
  done
done
