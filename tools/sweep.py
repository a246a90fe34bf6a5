#!/usr/bin/env python3
"""tools/sweep.py -- per-iteration time of every solver path over configs and batch sizes (GPU box)."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa
import flame_amd
from flame_amd import synth
from flame_amd.regularizer import OPT_PERSISTENT, OPT_DUAL_PUBLISH, RUN_PATHS

params = flame_amd.Params()
cases = [("640x480", 1), ("1280x720", 1), ("1920x1080", 1), ("640x480", 4), ("640x480", 7), ("640x480", 12), ("640x480", 15), ("640x480", 30)]
if len(sys.argv) > 1:
    cases = [(c.split(":")[0], int(c.split(":")[1])) for c in sys.argv[1:]]
for cfg, nf in cases:
    frames = [synth.make_graph(cfg, seed=5000 + i) for i in range(nf)]
    g = synth.concat_graphs(frames) if nf > 1 else frames[0]
    row = {}
    for form, dual, lds in ((4, 2, 0), (6, 2, 0), (3, 2, 0), (0, 0, 0)):  # patch per wave (one / two half-edges per lane), vertex per lane, one launch per step
        r = flame_amd.Regularizer(0)
        r.set_option(OPT_PERSISTENT, form)
        r.set_option(OPT_DUAL_PUBLISH, dual)
        r.upload_graph(g)
        iters = 200 if form else 50
        try:
            r.run(params, iters)
            ms = min(r.run_timed(params, iters) for _ in range(4))
            info = r.info()
            path = RUN_PATHS[info["last_run_path"]]
            us = ms * 1e3 / iters
            gbps = info["algorithmic_bytes_per_iter"] / (us * 1e-6) / 1e9
            row[path + ("+L2" if dual else "")] = f"{us:7.2f} us/it {nf / (us * 1e-6) / 1e6:6.2f} Mfi/s frac {gbps / 8000:5.3f}"
        except Exception as e:
            row[f"form{form}{lds}"] = f"ERR {e}"
        r.close()
    print(cfg, "x", nf, "V", g["V"], json.dumps(row))
