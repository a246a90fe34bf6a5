#!/bin/bash
# tools/profile.sh ROUND -- rocprofv3 evidence for bench.py's numbers (run on the GPU box via gpurun).
# Kernel trace/stats and the PMC counters are collected in SEPARATE runs (counters never together with
# tracing), as /opt/skills/guides prescribe.  Output: gpurun_out/prof_<ROUND>/ ; summarise with
# tools/summarize_profile.py and commit the summaries under profiles/.
set -u
R=${1:-r01}
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/prof_$R
mkdir -p $OUT
B="python bench.py --steps 20 --warmup 2 --no-extras --no-cpu-baseline"
rocprofv3 --kernel-trace --stats -d $OUT/kt -o kt --output-format csv -- $B > $OUT/kt.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d $OUT/fetch -o f --output-format csv -- $B > $OUT/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $OUT/write -o w --output-format csv -- $B > $OUT/write.log 2>&1
rocprofv3 --pmc SQ_WAVES GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY -d $OUT/sq -o s --output-format csv -- $B > $OUT/sq.log 2>&1
# the one-launch-per-step path, for comparison
B2="$B --persistent 0"
rocprofv3 --kernel-trace --stats -d $OUT/kt_step -o kt --output-format csv -- $B2 > $OUT/kt_step.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d $OUT/fetch_step -o f --output-format csv -- $B2 > $OUT/fetch_step.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $OUT/write_step -o w --output-format csv -- $B2 > $OUT/write_step.log 2>&1
# the HBM-streaming regime: 64 frames through k_fused_step (persistent forms off)
S="python tools/stream_case.py 64"
rocprofv3 --kernel-trace --stats -d $OUT/kt_stream -o kt --output-format csv -- $S > $OUT/kt_stream.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d $OUT/fetch_stream -o f --output-format csv -- $S > $OUT/fetch_stream.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $OUT/write_stream -o w --output-format csv -- $S > $OUT/write_stream.log 2>&1
# the resident batch (k_persistent_tv, 30 frames in one launch) and the per-feature epipolar update
rocprofv3 --kernel-trace --stats -d $OUT/kt_batch -o kt --output-format csv -- python tools/stream_case.py 30 resident > $OUT/kt_batch.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d $OUT/fetch_batch -o f --output-format csv -- python tools/stream_case.py 30 resident > $OUT/fetch_batch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $OUT/write_batch -o w --output-format csv -- python tools/stream_case.py 30 resident > $OUT/write_batch.log 2>&1
rocprofv3 --kernel-trace --stats -d $OUT/kt_stereo -o kt --output-format csv -- python tools/stereo_bench.py > $OUT/kt_stereo.log 2>&1
tail -1 $OUT/kt_batch.log; tail -3 $OUT/kt_stereo.log
grep -h '"metric"' $OUT/kt.log $OUT/kt_step.log | cut -c1-400
tail -1 $OUT/kt_stream.log
