"""tools/open_run_rate.py [SIZE] -- the open run's kernel instance against the plain one: 4 000 iterations in one launch each way (an open run left
to reach its bound), HIP events on the solver's stream, alternating; and how long an open run takes to stop once asked (sync() right after
run_open, by iterations done and by wall time).  GPU box."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import flame_amd
from flame_amd import synth

size = sys.argv[1] if len(sys.argv) > 1 else "640x480"
g = synth.make_graph(size, seed=1)
P = flame_amd.Params()
stream = torch.cuda.Stream(priority=-1)
with flame_amd.Regularizer(0) as reg:
    reg.set_stream(stream.cuda_stream)
    reg.upload_graph(g)
    reg.run(P, 400)
    if not reg.run_open(P, 64):
        print(size, ": an open run is not applicable here")
        sys.exit(0)
    reg.sync()
    t = {"plain": [], "open": []}
    for rep in range(12):
        for kind in ("plain", "open"):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            if kind == "plain":
                reg.run_async(P, 4000)
            else:
                assert reg.run_open(P, 4000)
            e1.record(stream)
            torch.cuda.synchronize()  # (not reg.sync(): the open run is to reach its bound)
            reg.sync()
            t[kind].append(e0.elapsed_time(e1) * 1e3 / 4000)
    for kind in t:
        v = np.array(t[kind][2:])
        print("%s %-5s: %.4f us per iteration (median of %d launches of 4000; min %.4f max %.4f)" % (size, kind, np.median(v), len(v), v.min(), v.max()))
    its, wall = [], []
    for rep in range(30):
        before = reg.iterations()[0]
        assert reg.run_open(P, 400000)
        time.sleep(0.0005)
        t0 = time.perf_counter()
        reg.sync()
        wall.append((time.perf_counter() - t0) * 1e6)
        its.append(reg.iterations()[0] - before)
    print("%s: sync() 0.5 ms after run_open returns after %.0f us (median; min %.0f, max %.0f); iterations done %d..%d" % (size, np.median(wall), min(wall), max(wall), min(its), max(its)))
