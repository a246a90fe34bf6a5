#!/usr/bin/env python3
"""tools/stream_case.py N [resident] -- an N-frame batch through the one-launch-per-step sweep (k_fused_step,
persistent forms off: the HBM-streaming regime of the solver) or, with `resident`, through the default path
(k_persistent_tv); profiled by tools/profile.sh."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa
import flame_amd
from flame_amd import synth
from flame_amd.regularizer import OPT_PERSISTENT

nf = int(sys.argv[1]) if len(sys.argv) > 1 else 64
resident = len(sys.argv) > 2 and sys.argv[2] == "resident"  # default run path (persistent, vertex-per-lane form)
g = synth.concat_graphs([synth.make_graph("640x480", seed=5000 + i) for i in range(nf)])
r = flame_amd.Regularizer(0)
if not resident:
    r.set_option(OPT_PERSISTENT, 0)
r.upload_graph(g)
p = flame_amd.Params()
r.run(p, 50)
ms = min(r.run_timed(p, 100) for _ in range(3))
info = r.info()
us = ms * 1e3 / 100
print(json.dumps({"frames": nf, "run_path": flame_amd.regularizer.RUN_PATHS.get(info["last_run_path"], "?"), "V": info["V"], "E": info["E"], "us_per_launch": round(us, 2),
                  "algorithmic_bytes_per_launch": info["algorithmic_bytes_per_iter"],
                  "algorithmic_GBps": round(info["algorithmic_bytes_per_iter"] / (us * 1e-6) / 1e9, 1)}))
