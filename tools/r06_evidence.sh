#!/bin/bash
# tools/r06_evidence.sh -- the round's measurements outside rocprofv3, on the GPU box (gpurun): every file lands in gpurun_out/r06/ and is
# copied under profiles/ by hand.  PARTS="bench loops cold soak" selects.  (The hand-off account has its own scripts: tools/r06_net_bench.sh,
# _net_bench2.sh, _net_trace.sh; the exit crash: tools/r06_segv*.sh; the C++ frame loop's gaps: tools/r06_frame_loop_gaps.sh.)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r06
mkdir -p $O
want() { [ -z "${PARTS:-}" ] || [[ " $PARTS " == *" $1 "* ]]; }
run() {  # run FILE command...: the command line, then its output
  local f=$1; shift
  echo "== $*" >> $O/$f
  timeout 600 "$@" >> $O/$f 2>> $O/$f.err
  echo >> $O/$f
}
if want bench; then
  python bench.py > $O/bench_1gpu.json 2> $O/bench_1gpu.err
  python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29521 bench.py --gpus 1 2> $O/bench_torchrun.err | grep '"metric"' > $O/bench_torchrun_1rank.json
  FLAME_BENCH_BACKEND=gloo FLAME_BENCH_DEVICE=0 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29522 bench.py --gpus 2 --steps 20 --warmup 3 2> $O/bench_2ranks.err | grep '"metric"' > $O/bench_torchrun_2ranks_gloo_one_device.json
fi
if want loops; then
  for s in 640x480 1280x720 1920x1080; do
    rm -f $O/frame_loop_$s.txt
    run frame_loop_$s.txt python tools/frame_loop.py --size $s --frames 12
    run frame_loop_$s.txt python tools/frame_loop.py --size $s --frames 20 --pipelined
  done
  rm -f $O/cpp_frame_loop_sizes.txt
  for s in 640x480 1920x1080; do run cpp_frame_loop_sizes.txt python tools/cpp_frame_loop.py --size $s --keep $O/keep; done
  rm -rf $O/keep
fi
if want cold; then
  rm -f $O/cold_start.txt $O/replay_rung.txt
  for s in 640x480 1920x1080; do run cold_start.txt python tools/cold_start.py $s; done
  for s in 640x480 1920x1080; do run replay_rung.txt python tools/replay_cost.py $s; done
fi
if want soak; then
  rm -f $O/soak_pipeline.txt $O/soak_under_load.txt
  run soak_pipeline.txt python tools/soak_pipeline.py 3000
  run soak_pipeline_1080p.txt env SOAK_FORM=1 python tools/soak_pipeline.py 2000 1920x1080
  run soak_under_load.txt python tools/soak.py 20000 load
fi
find $O -name "*.err" -size 0 -delete
ls -la $O
