"""tools/overlap_probe.py -- does IdepthGather.check_overlap see what the step loop then does?  One RCCL rank; the process's hardware-queue
count from the environment (GPU_MAX_HW_QUEUES, default of the runtime: 4).  Prints check_overlap's figures, then the step loop of bench.py
(set_export_target -> events -> run_async -> gather(async)) at several depths."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29534")
import torch, torch.distributed as dist
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
import flame_amd
from flame_amd import synth
from flame_amd.frames import IdepthGather
g = synth.make_graph("640x480", seed=1234)
dev = torch.device("cuda", 0)
p = flame_amd.Params()
reg = flame_amd.Regularizer(0)
if os.environ.get("PROBE_EARLY_RUN", "1") == "1":  # bench.py: the first frame of the process runs before the solver's torch stream exists
    reg.upload_graph(g); reg.run(p, 200)
stream = torch.cuda.Stream(device=dev, priority=-1)
reg.set_stream(stream.cuda_stream)
reg.upload_graph(g)
ig = IdepthGather(dist, [g["V"]], 1, dev, stream=None if os.environ.get("GATHER_NO_STREAM") else stream)
print("GPU_MAX_HW_QUEUES", os.environ.get("GPU_MAX_HW_QUEUES", "(runtime default)"), flush=True)
with torch.cuda.stream(stream):
    print("check_overlap:", ig.check_overlap(reg, p), flush=True)
if os.environ.get("PROBE_REUPLOAD", "1") == "1":
    reg.upload_graph(g)

def loop(steps, events, gather=True):
    reg.sync(); torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
    t0 = time.perf_counter()
    for e0, e1 in ev:
        if gather: reg.set_export_target(ig.local_row(0).data_ptr(), 1.0)
        if events: e0.record(stream)
        reg.run_async(p, 200)
        if events: e1.record(stream)
        if gather:
            with torch.cuda.stream(stream):
                ig.gather(async_op=True, regs=[reg])
    reg.sync(); ig.wait(); torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3

for steps in (24, 50, 200):
    for events in (False, True):
        loop(steps, events)
        print(f"steps {steps:3d} events {int(events)}: with gather {min(loop(steps, events) for _ in range(3)):.4f}  alone {min(loop(steps, events, False) for _ in range(3)):.4f} ms per step", flush=True)
with torch.cuda.stream(stream):
    print("check_overlap again:", ig.check_overlap(reg, p), flush=True)
reg.close(); dist.destroy_process_group()
