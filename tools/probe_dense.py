"""tools/probe_dense.py CASE.. -- in-kernel probe of k_persistent_pv by load: wait / compute cycles over the patches (percentiles), period,
poll rounds; PV_VARIANTS="113=1;108=9,113=4" = option settings to compare (GPU box).  CASE = SIZE:FRAMES, e.g. 640x480:4."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch  # noqa
import flame_amd
from flame_amd import synth
from flame_amd.regularizer import OPT_PERSISTENT, OPT_PROBE
params = flame_amd.Params()
N = 200
def probe(g, opts):
    r = flame_amd.Regularizer(0)
    try:
        for k, v in opts: r.set_option(k, v)
        r.set_option(OPT_PROBE, 1)
        r.upload_graph(g)
        r.run(params, N)
        ms = r.run_timed(params, N)
        p = r.read_probe().reshape(-1, N, 8).astype(np.int64)[:, 20:, :]
        p = p[p[:, 0, 5] != 0]
        info = r.info()
    finally:
        r.close()
    wait, comp = p[:, :, 2].mean(axis=1), p[:, :, 3].mean(axis=1)
    pct = lambda a: [round(float(np.percentile(a, q))) for q in (0, 5, 25, 50, 75, 95, 100)]
    dc = np.diff(p[0, :, 5]) & 0xffffffff
    dt = np.diff(p[0, :, 6]) & 0xffffffff
    # phase of every patch within the period: start of step 100 relative to patch 0 (mod period)
    period = float(dc.mean())
    hw = p[:, 0, 0]
    return {"path": info["last_run_path"], "patches": int(p.shape[0]), "us_per_iter": round(ms * 1e3 / N, 3), "period_cycles": round(period),
            "period_us": round(float(dt.mean()) / 100.0, 3), "GHz": round(period / (float(dt.mean()) * 10.0), 3),
            "wait_pct": pct(wait), "compute_pct": pct(comp), "rounds_mean": round(float(p[:, :, 4].mean()), 2)}
for c in sys.argv[1:]:
    cfg, nf = c.split(":"); nf = int(nf)
    frames = [synth.make_graph(cfg, seed=1234 + i) for i in range(nf)]
    g = frames[0] if nf == 1 else synth.concat_graphs(frames)
    for spec in [s for s in os.environ.get("PV_VARIANTS", "").split(";")]:
        opts = [(OPT_PERSISTENT, 4)] + [tuple(int(t) for t in kv.split("=")) for kv in spec.split(",") if kv]
        print(c, spec, json.dumps(probe(g, opts)), flush=True)
