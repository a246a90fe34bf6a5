#!/bin/bash
# tools/r05_evidence.sh -- the round's measurements outside rocprofv3, on the GPU box (gpurun): every file lands in gpurun_out/r05/ and is
# copied under profiles/ by hand.  PARTS="rg bench loops cold misc" selects.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r05
mkdir -p $O
want() { [ -z "${PARTS:-}" ] || [[ " $PARTS " == *" $1 "* ]]; }
run() {  # run FILE command...: the command line, then its output
  local f=$1; shift
  echo "== $*" >> $O/$f
  timeout 300 "$@" >> $O/$f 2>> $O/$f.err
  echo >> $O/$f
}
if want rg; then
  rm -f $O/rg_forms.txt $O/rg_step_cost.txt
  run rg_forms.txt python tools/rg_try.py 640x480,1280x720,1920x1080 1,2,3,4 0
  run rg_step_cost.txt python tools/rg_step_cost.py
fi
if want bench; then
  python bench.py > $O/bench_1gpu.json 2> $O/bench_1gpu.err
  python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29521 bench.py --gpus 1 2> $O/bench_torchrun.err | grep '"metric"' > $O/bench_torchrun_1rank.json
  python bench.py --persistent 7 --no-extras --no-cpu-baseline > $O/bench_rg.json 2> $O/bench_rg.err
fi
if want loops; then
  for s in 640x480 1280x720 1920x1080; do
    rm -f $O/frame_loop_$s.txt
    run frame_loop_$s.txt python tools/frame_loop.py --size $s --frames 12 --cpu
    run frame_loop_$s.txt python tools/frame_loop.py --size $s --frames 20 --pipelined
  done
  rm -f $O/cpp_frame_loop.txt
  run cpp_frame_loop.txt python -m pytest tests/test_cpp_facade.py -q -m gpu -k frame_loop_end_to_end -s
fi
if want cold; then
  rm -f $O/cold_start.txt $O/replay_rung.txt
  for s in 640x480 1920x1080; do run cold_start.txt python tools/cold_start.py $s; done
  for s in 640x480 1920x1080; do run replay_rung.txt python tools/replay_cost.py $s; done
fi
if want misc; then
  rm -f $O/gather_tax.txt $O/delaunay.txt $O/gather_overlap.txt
  run gather_tax.txt python tools/export_tax.py
  run gather_tax.txt env FREE_CUS_PER_XCD=1 python tools/export_tax.py   # (the solver's stream leaves one / two compute units per XCD to the collective)
  run gather_tax.txt env FREE_CUS_PER_XCD=2 python tools/export_tax.py
  for q in 4 8 1; do run gather_overlap.txt env GPU_MAX_HW_QUEUES=$q python tools/overlap_probe.py; done
  run gather_tax.txt bash tools/gather_tax.sh
  run delaunay.txt python tools/delaunay_probe.py
fi
ls -la $O
