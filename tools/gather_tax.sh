set -x
B="bench.py --steps 200 --warmup 20 --no-extras --no-cpu-baseline"
T="python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1"
python $B 2>/dev/null | tail -1 | python -c "import json,sys; o=json.loads(sys.stdin.read()); print('plain', o['value'], o['ms_per_step'])"
$T --master-port 29511 $B --gpus 1 2>/dev/null | grep '"metric"' | python -c "import json,sys; o=json.loads(sys.stdin.read()); print('torchrun', o['value'], o['ms_per_step'], o['result_gather'])"
NCCL_MAX_NCHANNELS=1 NCCL_MIN_NCHANNELS=1 $T --master-port 29512 $B --gpus 1 2>/dev/null | grep '"metric"' | python -c "import json,sys; o=json.loads(sys.stdin.read()); print('torchrun 1 channel', o['value'], o['ms_per_step'], o['result_gather'])"
NCCL_MAX_NCHANNELS=1 NCCL_MIN_NCHANNELS=1 NCCL_NTHREADS=64 $T --master-port 29513 $B --gpus 1 2>/dev/null | grep '"metric"' | python -c "import json,sys; o=json.loads(sys.stdin.read()); print('torchrun 1 channel 64 threads', o['value'], o['ms_per_step'], o['result_gather'])"
