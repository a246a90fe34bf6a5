#!/bin/bash
# tools/r06_segv_bisect.sh -- which ingredient makes a process of this library die in the HIP runtime's tearDown under rocprofv3.  GPU box.
set -u
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/segv_bisect
mkdir -p $OUT
run() {  # run TAG [ENV=.. ...] -- python code
  local tag=$1; shift
  local envs=()
  while [ "$1" != "--" ]; do envs+=("$1"); shift; done
  shift
  env "${envs[@]}" timeout 200 rocprofv3 --kernel-trace -d $OUT/$tag -o kt --output-format csv -- "$@" > $OUT/$tag.log 2>&1
  local rc=$?
  printf "%-34s exit %3d  segv-lines %s\n" "$tag" $rc "$(grep -c 'SIGSEGV\|Segmentation' $OUT/$tag.log)"
}
PRE="import sys, os; sys.path.insert(0, os.getcwd()); import torch"
run t0_torch_only -- python -c "$PRE; x = torch.zeros(8, device='cuda'); torch.cuda.synchronize()"
run t1_torch_hiprio_stream -- python -c "$PRE; s = torch.cuda.Stream(priority=-1)
with torch.cuda.stream(s): x = torch.zeros(8, device='cuda') + 1
torch.cuda.synchronize()"
CREATE="$PRE; import flame_amd; r = flame_amd.Regularizer(0)"
run c0_create_close -- python -c "$CREATE; r.close()"
run c1_create_close_nowarm_lazy FLAME_NLTGV2_NO_WARM=1 FLAME_NLTGV2_LAZY_CALIBRATION=1 -- python -c "$CREATE; r.close()"
run c2_create_close_nowarm FLAME_NLTGV2_NO_WARM=1 -- python -c "$CREATE; r.close()"
run c3_create_close_lazy FLAME_NLTGV2_LAZY_CALIBRATION=1 -- python -c "$CREATE; r.close()"
run c4_create_noclose -- python -c "$CREATE"
RUN="$CREATE; from flame_amd import synth; g = synth.make_graph('320x240', seed=1); r.upload_graph(g); r.run(flame_amd.Params(), 50)"
run c5_run_close -- python -c "$RUN; r.close()"
run c6_run_close_nowarm_lazy FLAME_NLTGV2_NO_WARM=1 FLAME_NLTGV2_LAZY_CALIBRATION=1 -- python -c "$RUN; r.close()"
run c7_run_noplacement -- python -c "$CREATE; from flame_amd import synth; from flame_amd.regularizer import OPT_PLACEMENT; r.set_option(OPT_PLACEMENT, 0); g = synth.make_graph('320x240', seed=1); r.upload_graph(g); r.run(flame_amd.Params(), 50); r.close()"
run c8_run_perstep -- python -c "$CREATE; from flame_amd import synth; from flame_amd.regularizer import OPT_PERSISTENT; r.set_option(OPT_PERSISTENT, 0); g = synth.make_graph('320x240', seed=1); r.upload_graph(g); r.run(flame_amd.Params(), 50); r.close()"
# no torch, no python: the C++ facade test
g++ -std=c++11 -O1 -I include tests/cpp/facade_test.cc -o $OUT/facade_test -L flame_amd -lflame_nltgv2_hip -L oracle -loracle_nltgv2 \
  -Wl,-rpath,$PWD/flame_amd -Wl,-rpath,$PWD/oracle -Wl,-rpath,/opt/rocm/lib 2> $OUT/facade_build.log
run f0_cpp_facade -- $OUT/facade_test
run f1_cpp_facade_nowarm_lazy FLAME_NLTGV2_NO_WARM=1 FLAME_NLTGV2_LAZY_CALIBRATION=1 -- $OUT/facade_test
