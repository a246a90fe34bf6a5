#!/bin/bash
# tools/profile.sh ROUND -- rocprofv3 evidence for bench.py's numbers (run on the GPU box via gpurun).
# Kernel trace/stats and the PMC counters are collected in SEPARATE runs (counters never together with
# tracing), as /opt/skills/guides prescribe.  Output: gpurun_out/prof_<ROUND>/ ; summarise with
# tools/summarize_profile.py and commit the summaries under profiles/.
set -u
R=${1:-r06}
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/prof_$R
mkdir -p $OUT
want() { [ -z "${CASES:-}" ] || [[ " $CASES " == *" $1 "* ]]; }  # CASES="cfg5 batch10": only those
prof() {  # prof TAG "command": kernel stats + FETCH + WRITE + SQ counters, each in its own process
  local tag=$1; shift
  want $tag || return 0
  timeout 150 rocprofv3 --kernel-trace --stats -d $OUT/kt_$tag -o kt --output-format csv -- "$@" > $OUT/kt_$tag.log 2>&1
  if [ -n "${KT_ONLY:-}" ]; then tail -1 $OUT/kt_$tag.log | cut -c1-400; return 0; fi   # KT_ONLY=1: the kernel statistics alone (the counters are kept)
  timeout 150 rocprofv3 --pmc FETCH_SIZE -d $OUT/fetch_$tag -o f --output-format csv -- "$@" > $OUT/fetch_$tag.log 2>&1
  timeout 150 rocprofv3 --pmc WRITE_SIZE -d $OUT/write_$tag -o w --output-format csv -- "$@" > $OUT/write_$tag.log 2>&1
  timeout 150 rocprofv3 --pmc SQ_WAVES GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY -d $OUT/sq_$tag -o s --output-format csv -- "$@" > $OUT/sq_$tag.log 2>&1
  tail -1 $OUT/kt_$tag.log | cut -c1-400
}
prof_onchip() {  # the on-chip side of a resident (batched) kernel: LDS and instruction-class activity, two more counter passes
  local tag=$1; shift
  want $tag || return 0
  [ -z "${KT_ONLY:-}" ] || return 0
  timeout 150 rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES -d $OUT/lds_$tag -o l --output-format csv -- "$@" > $OUT/lds_$tag.log 2>&1
  timeout 150 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM_RD SQ_INSTS_VMEM_RD SQ_INSTS_SALU -d $OUT/act_$tag -o a --output-format csv -- "$@" > $OUT/act_$tag.log 2>&1
}
B="python bench.py --steps 20 --warmup 2 --no-extras --no-cpu-baseline"
prof bench $B                                   # the bench line's own command: 640x480, default path (k_persistent_pv)
prof bench_tv $B --persistent 3                 # the vertex-per-lane kernel on the same workload
prof bench_step $B --persistent 0               # one launch per step
prof cfg3 python tools/profile_case.py single:1280x720
prof cfg5 python tools/profile_case.py single:1920x1080
prof batch5 python tools/profile_case.py batch:5:200     # five frames in the patch-per-wave kernel (19 patches per CU; round 3's planner)
prof batch10 python tools/profile_case.py batch:10:200   # bench.py batched.ten: ten frames in ONE launch of k_persistent_pv2
prof batch30 python tools/profile_case.py batch:30:200   # bench.py batched.resident (200 iterations per launch)
prof batch64 python tools/profile_case.py batch:64:100   # bench.py batched.large (100 iterations, 3 launch groups)
prof stream64 python tools/profile_case.py stream:64
prof_onchip cfg5 python tools/profile_case.py single:1920x1080
prof_onchip batch10 python tools/profile_case.py batch:10:200
prof_onchip batch30 python tools/profile_case.py batch:30:200
prof_onchip batch64 python tools/profile_case.py batch:64:100
want stereo || exit 0
timeout 150 rocprofv3 --kernel-trace --stats -d $OUT/kt_stereo -o kt --output-format csv -- python tools/stereo_bench.py > $OUT/kt_stereo.log 2>&1
[ -z "${KT_ONLY:-}" ] && timeout 150 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_INSTS_LDS -d $OUT/sq_stereo -o s --output-format csv -- python tools/stereo_bench.py > $OUT/sq_stereo.log 2>&1
tail -3 $OUT/kt_stereo.log
