"""Cost of switching the standing export target between chained asynchronous runs (the double-buffered rows of a result gather)."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import flame_amd
from flame_amd import synth

g = synth.make_graph("640x480", seed=1234)
P = flame_amd.Params()
reg = flame_amd.Regularizer(0)
st = torch.cuda.Stream(priority=-1)
reg.set_stream(st.cuda_stream)
reg.upload_graph(g)
rows = [torch.zeros(g["V"], device="cuda"), torch.zeros(g["V"], device="cuda")]
reg.run(P, 200)
for mode in ("no target", "one target", "two targets alternating"):
    for rep in range(3):
        reg.sync(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(50):
            if mode == "one target":
                reg.set_export_target(rows[0].data_ptr(), 1.0)
            elif mode.startswith("two"):
                reg.set_export_target(rows[k & 1].data_ptr(), 1.0)
            reg.run_async(P, 200)
        t1 = time.perf_counter()
        reg.sync(); torch.cuda.synchronize()
        t2 = time.perf_counter()
    print("%-26s enqueue %.3f ms per step, total %.3f ms per step" % (mode, (t1 - t0) * 20, (t2 - t0) * 20))
    reg.set_export_target(0, 1.0)
