#!/bin/bash
# lone-wave pace and headline frame for builds given as file names under lib/
L=$PWD/light-path-tracer_amd/lib
for lib in "$@"; do
  echo "== $lib"
  LTRACE_LIB=$L/$lib LT_STAMPS_FILE=/tmp/s_$$.bin python tools/lone_pace_by_lanes.py 2>&1 | tail -1
  LTRACE_LIB=$L/$lib python tests/ghost_lanes_check.py | grep digest
done
bash tools/scratch/ab.sh "$1" "$2"
bash tools/scratch/ab.sh "$1" "$2" --size 2048
