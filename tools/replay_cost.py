"""tools/replay_cost.py [SIZE] -- what an expired chain costs: a chain of 4 x 200 iterations clean / with an injected fault (taken back, redone
persistently at reduced residency) / with a fault that also hits the replay (redone one launch per step), wall time around sync()."""
import os, sys, time
import numpy as np
import torch  # noqa: F401
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import flame_amd
from flame_amd import synth
from flame_amd.regularizer import OPT_FAULT_INJECT, RUN_PATHS
size = sys.argv[1] if len(sys.argv) > 1 else "1920x1080"
g = synth.make_graph(size, seed=3)
p = flame_amd.Params()
def chain(reg):
    reg.sync()
    t0 = time.perf_counter()
    for _ in range(4):
        reg.run_async(p, 200)
    reg.sync()
    return (time.perf_counter() - t0) * 1e3
for name, fault in (("clean", 0), ("fault -> persistent replay", 64), ("fault in the replay too -> per-step replay", (1 << 22) + 64)):
    ts, paths = [], []
    for rep in range(5):
        with flame_amd.Regularizer(0) as reg:
            reg.upload_graph(g)
            chain(reg)
            reg.set_option(OPT_FAULT_INJECT, fault)
            ts.append(chain(reg))
            paths.append(RUN_PATHS[reg.info()["last_run_path"]])
    print(f"{size} {name}: chain of 4 x 200 iterations {np.median(ts):.3f} ms (min {min(ts):.3f}), last path {paths[-1]}", flush=True)
