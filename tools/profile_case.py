#!/usr/bin/env python3
"""tools/profile_case.py CASE -- one workload per process, for rocprofv3 (tools/profile.sh).

  single:<config>[:<persistent option>]   one frame, 200-iteration runs (1920x1080 with the fused photometric residual)
  batch:<frames>[:<iters>]                frames of 640x480 as one disjoint union, default path (resident: k_persistent_tv);
                                          iterations per run as bench.py's batched lines use them (resident 200, large 100)
  stream:<frames>[:<iters>]               the same through the one-launch-per-step sweep (k_fused_step)
Prints one JSON line with the HIP-event time per launch and the algorithmic bytes per launch."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch  # noqa: F401

import flame_amd
from flame_amd import synth
from flame_amd.regularizer import OPT_PERSISTENT, RUN_PATHS

case = sys.argv[1] if len(sys.argv) > 1 else "single:640x480"
kind, _, rest = case.partition(":")
p = flame_amd.Params()
r = flame_amd.Regularizer(0)
iters = 200
if kind == "single":
    cfg, _, opt = rest.partition(":")
    if opt:
        r.set_option(OPT_PERSISTENT, int(opt))
    g = synth.make_graph(cfg, seed=1234)
    r.upload_graph(g)
    if cfg == "1920x1080":  # BASELINE config 5: the residual of the final x comes out of the solver's own launch
        from flame_amd import synth_stereo as ss

        w, h, _ = synth.CONFIGS[cfg]
        ref_img = np.clip(np.rint(ss.texture(w, h, 77, margin=0)), 0, 255).astype(np.uint8)
        K = np.array([[0.52 * w, 0, w / 2.0], [0, 0.52 * w, h / 2.0], [0, 0, 1]])
        r.photo_set_images(ref_img, np.roll(ref_img, 3, axis=1))
        r.photo_fuse(np.eye(3, dtype=np.float32), (K @ np.array([0.04, -0.01, 0.003])).astype(np.float32), graph_scale=1.0, border=4)
else:
    nfs, _, its = rest.partition(":")
    nf = int(nfs)
    g = synth.concat_graphs([synth.make_graph("640x480", seed=5000 + i) for i in range(nf)])
    if kind == "stream":
        r.set_option(OPT_PERSISTENT, 0)
    iters = int(its) if its else 100
    r.upload_graph(g)
r.run(p, iters)
ms = min(r.run_timed(p, iters) for _ in range(5))
info = r.info()
path = RUN_PATHS.get(info["last_run_path"], "?")
launches = 1 if path.startswith("persistent") else iters
groups = max(1, info["last_run_groups"]) if path.startswith("persistent") else 1
print(json.dumps({"case": case, "run_path": path, "V": info["V"], "E": info["E"], "iters_per_run": iters,
                  "launches_per_run": launches * groups, "us_per_run": round(ms * 1e3, 2), "us_per_iter": round(ms * 1e3 / iters, 3),
                  "algorithmic_bytes_per_iter": info["algorithmic_bytes_per_iter"],
                  "algorithmic_GBps": round(info["algorithmic_bytes_per_iter"] * iters / (ms * 1e-3) / 1e9, 1)}))
r.close()
