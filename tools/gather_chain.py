"""Where a step's time goes with the result gather in the loop (one rank, RCCL): host time of each call, device time of each run."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import torch.distributed as dist
import flame_amd
from flame_amd import synth
from flame_amd.frames import IdepthGather

os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29517")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
torch.cuda.set_device(0)
if os.environ.get("GC_NO_DEVICE_ID"):
    dist.init_process_group(backend="nccl")
else:
    dist.init_process_group(backend="nccl", device_id=torch.device("cuda", 0))
g = synth.make_graph("640x480", seed=1234)
P = flame_amd.Params()
reg = flame_amd.Regularizer(0)
st = torch.cuda.Stream(priority=-1)
reg.set_stream(st.cuda_stream)
reg.upload_graph(g)
ig = IdepthGather(dist, [g["V"]], 1, torch.device("cuda", 0))
for mode in ("no gather", "gather async", "gather async, no regs"):
    for rep in range(3):
        reg.sync(); torch.cuda.synchronize()
        tt = [0.0, 0.0, 0.0]
        ev = []
        t0 = time.perf_counter()
        for k in range(50):
            a = time.perf_counter()
            reg.set_export_target(ig.local_row(0).data_ptr(), 1.0)
            b = time.perf_counter()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(st); reg.run_async(P, 200); e1.record(st)
            c = time.perf_counter()
            if mode != "no gather":
                with torch.cuda.stream(st):
                    ig.gather(async_op=True, regs=[reg] if mode == "gather async" else None)
            d = time.perf_counter()
            tt[0] += b - a; tt[1] += c - b; tt[2] += d - c
            ev.append((e0, e1))
        reg.sync(); torch.cuda.synchronize()
        t2 = time.perf_counter()
        ig.wait()
    dev = sum(e0.elapsed_time(e1) for e0, e1 in ev) / 50
    print("%-24s wall %.3f ms per step; host: target %.3f, run_async %.3f, gather %.3f ms; device run %.3f ms" % (mode, (t2 - t0) * 20, tt[0] * 20, tt[1] * 20, tt[2] * 20, dev))
dist.destroy_process_group()
