#!/bin/bash
# tools/r06_early_expand.sh -- sync_commit with the next topology's expansion enqueued beside the solver's last rounds, against the settle-first
# order (FLAME_NLTGV2_LATE_EXPAND=1): the whole GPU suite, then both frame loops either way.  GPU box.
set -u
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/early_expand
mkdir -p $OUT
timeout 1500 python -m pytest tests -q -m gpu -x > $OUT/tests.log 2>&1
tail -3 $OUT/tests.log
python tools/cpp_frame_loop.py --host-work-us 300 --keep $OUT/keep640 > $OUT/cpp640.log 2>&1
for rep in 1 2 3; do
  echo "== C++ loop, expansion beside the solver, run $rep"
  $OUT/keep640/frame_loop_test $OUT/keep640/frames.bin $OUT/log_lean.bin 200 1 2>&1 | grep -A2 "frame loop" | cut -c1-420
  echo "== C++ loop, FLAME_NLTGV2_LATE_EXPAND=1, run $rep"
  FLAME_NLTGV2_LATE_EXPAND=1 $OUT/keep640/frame_loop_test $OUT/keep640/frames.bin $OUT/log_lean.bin 200 1 2>&1 | grep -A2 "frame loop" | cut -c1-420
done
FLAME_NLTGV2_TRACE=1 $OUT/keep640/frame_loop_test $OUT/keep640/frames.bin $OUT/log_lean.bin 200 1 2>&1 | grep "sync_graph (device)" | tail -4 | cut -c1-420
for rep in 1 2; do
  echo "== tools/frame_loop.py --pipelined, expansion beside the solver, run $rep"
  python tools/frame_loop.py --pipelined --frames 20 2>&1 | tail -3 | cut -c1-420
  echo "== tools/frame_loop.py --pipelined, FLAME_NLTGV2_LATE_EXPAND=1, run $rep"
  FLAME_NLTGV2_LATE_EXPAND=1 python tools/frame_loop.py --pipelined --frames 20 2>&1 | tail -3 | cut -c1-420
done
rm -rf $OUT/keep640 $OUT/log_lean.bin
