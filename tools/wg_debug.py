#!/usr/bin/env python3
"""tools/wg_debug.py -- pv form: rounds of local-only polling before the remote records are polled too (GPU box)."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch  # noqa
import flame_amd
from flame_amd import synth
from tools.wg_sweep import timed, probe_summary
from flame_amd.regularizer import OPT_PERSISTENT, OPT_WG_WAVES, OPT_POLL_GAP, OPT_PRESLEEP, OPT_XCDS, OPT_PW_ROLES, OPT_DUAL_PUBLISH

for cfg in ("320x240", "640x480", "1280x720"):
    g = synth.make_graph(cfg, seed=5000)
    us_he, path, want, _ = timed(g, [(OPT_PERSISTENT, 2)])
    row = {"cfg": cfg, "he": round(us_he, 3)}
    for gap in (1, 2):
        for ps in (1, 3, 5, 7, 9, 13):
            opts = [(OPT_PERSISTENT, 4), (OPT_WG_WAVES, 1), (OPT_PW_ROLES, 0), (OPT_POLL_GAP, gap), (OPT_PRESLEEP, ps)]
            us, path, out, same = timed(g, opts, want=want)
            row[f"gap{gap} lr{ps - 1}"] = round(us, 3) if (path == "persistent-wg" and same) else f"{path} {same}"
    print(json.dumps(row), flush=True)
