"""tools/cold_start.py [SIZE] -- what the first frame of a process and of a fresh context costs: create, upload_graph, the first run (sync)."""
import os, sys, time
import numpy as np
import torch  # noqa: F401
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import flame_amd
from flame_amd import synth
size = sys.argv[1] if len(sys.argv) > 1 else "640x480"
g = synth.make_graph(size, seed=1)
p = flame_amd.Params()
torch.cuda.synchronize()
for ctx_no in range(3):
    t0 = time.perf_counter(); reg = flame_amd.Regularizer(0); t1 = time.perf_counter()
    reg.upload_graph(g); reg.sync(); t2 = time.perf_counter()
    reg.run(p, 200); t3 = time.perf_counter()
    reg.run(p, 200); t4 = time.perf_counter()
    reg.upload_graph(g); reg.sync(); t5 = time.perf_counter()
    reg.run(p, 200); t6 = time.perf_counter()
    print(f"{size} context {ctx_no}: create {1e3 * (t1 - t0):.3f} ms, upload_graph {1e3 * (t2 - t1):.3f}, first run of 200 {1e3 * (t3 - t2):.3f}, second run {1e3 * (t4 - t3):.3f}, "
          f"re-upload {1e3 * (t5 - t4):.3f}, run after it {1e3 * (t6 - t5):.3f}  -> first frame {1e3 * (t3 - t1):.3f} ms, a graph reset {1e3 * (t6 - t4):.3f} ms", flush=True)
    reg.close()
