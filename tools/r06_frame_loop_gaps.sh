#!/bin/bash
# tools/r06_frame_loop_gaps.sh -- kernel trace of the C++ frame loop (lean mode) and what fills the gaps between solver launches.  GPU box.
set -u
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/frame_loop_gaps
mkdir -p $OUT
FLAME_KEEP_FRAME_LOOP=$PWD/$OUT/keep timeout 600 python -m pytest tests/test_cpp_facade.py -q -m gpu -k frame_loop_end_to_end -s > $OUT/test.log 2>&1
grep "frame loop" $OUT/test.log
python tools/cpp_frame_loop.py --keep $OUT/keep640 > $OUT/size640.log 2>&1; cat $OUT/size640.log
timeout 300 rocprofv3 --kernel-trace -d $OUT/kt -o kt --output-format csv -- $OUT/keep640/frame_loop_test $OUT/keep640/frames.bin $OUT/log_lean.bin 200 1 > $OUT/prof.log 2>&1
echo "profiled run: exit $?"; grep "frame loop" $OUT/prof.log
FLAME_NLTGV2_TRACE=1 $OUT/keep640/frame_loop_test $OUT/keep640/frames.bin $OUT/log_lean.bin 200 1 2>&1 | grep "interpolate_mesh_begin\|sync_graph (device)" | tail -8
python tools/r06_frame_loop_gaps.py $(find $OUT/kt -name "*kernel_trace.csv" | head -1)
rm -rf $OUT/keep $OUT/keep640 $OUT/log_lean.bin  # (inputs and logs of tens of MB: gpurun_out/ travels back only below 64 MiB)
