"""tools/launch_intercept_trace.py -- run under `rocprofv3 --kernel-trace`: 12 back-to-back run_async launches each of n = 8, 16, 50, 100, 200, 400, 800
iterations (640x480); tools/launch_intercept_fit.py reads the trace: the kernel's own duration by n (intercept = what a launch costs inside the
kernel: wave launch, rotation word, table loads, XCC table, epilogue) and the gap between two launches of the stream.  GPU box."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401

import flame_amd
from flame_amd import synth

g = synth.make_graph(sys.argv[1] if len(sys.argv) > 1 else "640x480", seed=1)
P = flame_amd.Params()
with flame_amd.Regularizer(0) as reg:
    reg.upload_graph(g)
    reg.run(P, 200)
    for n in (8, 16, 50, 100, 200, 400, 800):
        for _ in range(12):
            reg.run_async(P, n)
        reg.sync()
    print("run path:", flame_amd.regularizer.RUN_PATHS.get(reg.info()["last_run_path"]))
