"""tools/export_tax.py -- what the standing export target costs a step when it alternates between two rows (the double-buffered gather): run alone / + set_export_target / + RCCL gather."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch, torch.distributed as dist
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
import flame_amd
from flame_amd import synth
from flame_amd.frames import IdepthGather
g = synth.make_graph("640x480", seed=1234)
dev = torch.device("cuda", 0)
reg = flame_amd.Regularizer(0)
free = int(os.environ.get("FREE_CUS_PER_XCD", "0"))
if free:
    # the solver's stream may use all compute units but `free` per XCD (bit i of the mask = compute unit i, 32 per XCD):
    # the collective's kernel then finds compute units no patch lives on
    import ctypes
    hip = ctypes.CDLL("libamdhip64.so")
    words = (ctypes.c_uint32 * 8)(*([0xffffffff >> free] * 8))
    sp = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(sp), 8, words)
    assert rc == 0, rc
    stream = torch.cuda.ExternalStream(sp.value, device=dev)
    print(f"solver stream with a CU mask: {free} free per XCD", flush=True)
else:
    stream = torch.cuda.Stream(device=dev, priority=-1)
reg.set_stream(stream.cuda_stream)
reg.upload_graph(g)
ig = IdepthGather(dist, [g["V"]], 1, dev, stream=None if os.environ.get("GATHER_NO_STREAM") else stream)
p = flame_amd.Params()
rows = torch.zeros((2, g["V"]), dtype=torch.float32, device=dev)
def loop(mode, steps=200):
    reg.sync(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    with torch.cuda.stream(stream):
        for s in range(steps):
            if mode == 1: reg.set_export_target(rows[s & 1].data_ptr(), 1.0)
            if mode == 2: reg.set_export_target(rows[0].data_ptr(), 1.0)
            if mode == 3: reg.set_export_target(ig.local_row(0).data_ptr(), 1.0)
            reg.run_async(p, 200)
            if mode == 3: ig.gather(async_op=True, regs=[reg])
    reg.sync(); ig.wait(); torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3
for m, name in ((0, "run alone"), (2, "+ export, one row"), (1, "+ export, two rows alternating"), (3, "+ export + RCCL gather")):
    loop(m, 20)
    print(f"{name}: {min(loop(m) for _ in range(3)):.4f} ms per step", flush=True)
reg.close(); dist.destroy_process_group()
