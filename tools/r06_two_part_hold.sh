#!/bin/bash
# tools/r06_two_part_hold.sh -- the C++ and the Python frame loop with the mesh begun beside the next round (FLAME_NLTGV2_OPT_MESH_STATE = 1,
# SolverLoop::withDevice(f, g)) against the one-part hold.  GPU box.
set -u
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/two_part
mkdir -p $OUT
timeout 900 python -m pytest tests/test_sync_graph.py tests/test_cpp_facade.py -q -m gpu -x -k "mesh_of_the_state or frame_loop or resident_dense_map or integration_switch" > $OUT/tests.log 2>&1
tail -3 $OUT/tests.log
for hw in 300 0; do
  python tools/cpp_frame_loop.py --host-work-us $hw --keep $OUT/keep640 > $OUT/two_part_hw$hw.log 2>&1
  echo "== two-part hold, $hw us of other host work"; grep -A2 "lean" $OUT/two_part_hw$hw.log | cut -c1-420
  for rep in 1 2; do
    echo "== one-part hold (FRAME_LOOP_ONE_PART=1), $hw us, run $rep"
    FRAME_LOOP_ONE_PART=1 $OUT/keep640/frame_loop_test $OUT/keep640/frames.bin $OUT/log_lean.bin 200 1 2>&1 | grep -A2 "frame loop" | cut -c1-420
    echo "== two-part hold, $hw us, run $rep"
    $OUT/keep640/frame_loop_test $OUT/keep640/frames.bin $OUT/log_lean.bin 200 1 2>&1 | grep -A2 "frame loop" | cut -c1-420
  done
done
for ms in 1 0; do
  echo "== tools/frame_loop.py --pipelined --mesh-state $ms"
  python tools/frame_loop.py --pipelined --frames 20 --mesh-state $ms 2>&1 | tail -4 | cut -c1-420
done
rm -rf $OUT/keep640 $OUT/log_lean.bin
