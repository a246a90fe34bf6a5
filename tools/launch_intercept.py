"""tools/launch_intercept.py [SIZE] -- what ONE launch of the persistent solver costs beyond its iterations: run_timed (HIP events around the
launch on the solver's stream) for n = 50 ... 3200 iterations, a straight line through the medians; the intercept is the launch's fixed
cost (wave launch, prologue, first-iteration skew, epilogue), the slope the steady period.  And the same for TWO launches back to back
(run_async twice between one pair of events): the second launch's extra = fixed cost + the gap between two launches.  GPU box."""
import os
import sys

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
    ns = [50, 100, 200, 400, 800, 1600, 3200]
    med = []
    for n in ns:
        t = []
        for _ in range(25):
            t.append(reg.run_timed(P, n) * 1e3)
        med.append(float(np.median(t[3:])))
    A = np.vstack([np.ones(len(ns)), ns]).T
    (b, a), *_ = np.linalg.lstsq(A, np.array(med), rcond=None)
    print("%s: one launch, us by iterations: %s" % (size, ", ".join("%d: %.1f" % (n, m) for n, m in zip(ns, med))))
    print("   fit: %.2f us + %.4f us per iteration (the launch's fixed cost inside its own events; steady period)" % (b, a))
    for n in (100, 200, 400):
        t1, t2 = [], []
        for _ in range(25):
            e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
            e0.record(stream)
            reg.run_async(P, n)
            e1.record(stream)
            reg.run_async(P, n)
            e2.record(stream)
            reg.sync()
            torch.cuda.synchronize()
            t1.append(e0.elapsed_time(e1) * 1e3), t2.append(e1.elapsed_time(e2) * 1e3)
        print("   two launches of %d back to back: first %.1f us, second %.1f us (= %.1f us more than %d iterations at the steady period)"
              % (n, np.median(t1[3:]), np.median(t2[3:]), np.median(t2[3:]) - n * a, n))
    print("   run path:", flame_amd.regularizer.RUN_PATHS.get(reg.info()["last_run_path"]))
