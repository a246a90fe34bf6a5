#!/bin/bash
# tools/ab_builds.sh LIB_A LIB_B [CASES...] -- same-box A/B of two builds of the library (build/ab/libX.so: copies of flame_amd/libflame_nltgv2_hip.so
# made from two states of the tree), alternating, tools/time_cases.py with 2000 iterations per launch.  GPU box.
cd "$(dirname "$0")/.."
A=$1; B=$2; shift 2
CASES=${@:-"640x480:1 1280x720:1 320x240:1 1920x1080:1 640x480:3"}
for round in 1 2 3; do
  for L in $A $B; do
    echo "== $L (round $round)"
    FLAME_AMD_LIBRARY=$PWD/build/ab/lib$L.so TC_ITERS=2000 python tools/time_cases.py $CASES 2>/dev/null
  done
done
