#!/usr/bin/env python3
"""tools/wg_hop.py -- per-record hand-off latency inside k_persistent_wg (GPU box): every published record carries
its producer's 100 MHz clock; the communication wave of each consuming workgroup notes (its clock at detection -
that) for every record it fetched, every step.  Prints percentiles, same-XCD vs cross-XCD, and the per-workgroup
maximum per step (what the workgroup actually waits for)."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch  # noqa
import flame_amd
from flame_amd import synth
from flame_amd.regularizer import (OPT_PERSISTENT, OPT_WG_WAVES, OPT_POLL_GAP, OPT_PRESLEEP, OPT_WG_RECORD, OPT_PROBE, OPT_XCDS)

p = flame_amd.Params()
N = 200
cfg = sys.argv[1] if len(sys.argv) > 1 else "640x480"
g = synth.make_graph(cfg, seed=5000)
res = []
for W in (2, 4):
    for ps in (1, 25):
        for xc in ((0,) if g["V"] > 4000 else (1, 8)):
            r = flame_amd.Regularizer(0)
            for k, v in [(OPT_PERSISTENT, 4), (OPT_WG_WAVES, W), (OPT_POLL_GAP, 1), (OPT_PRESLEEP, ps), (OPT_WG_RECORD, 0x15), (OPT_PROBE, 1), (OPT_XCDS, xc)]:
                r.set_option(k, v)
            r.upload_graph(g)
            r.run(p, N)
            ms = r.run_timed(p, N)
            raw = r.read_probe()
            info = r.info()
            r.close()
            n_lat = raw.size // (N * ((W + 1) * 8 + 128))  # = workgroups
            w1 = n_lat * (W + 1) * N * 8
            pr = raw[:w1].reshape(n_lat, W + 1, N, 8).astype(np.int64)
            lat = raw[w1:].reshape(n_lat, N, 128)
            my_xcc = pr[:, W, 0, 3]                      # the communication wave's record, word 3
            ok = lat != 0xffffffff
            ok[:, :20, :] = False
            d = (lat & 0xffffff).astype(np.int64) * 10    # ns
            pxcc = (lat >> 24).astype(np.int64)
            same = ok & (pxcc == my_xcc[:, None, None])
            cross = ok & ~same
            def pct(a):
                return {q: round(float(np.percentile(a, q)), 0) for q in (5, 25, 50, 75, 95, 99)} if a.size else {}
            dd = np.where(ok, d, -1)
            wg_max = dd.max(axis=2)[:, 20:]               # per workgroup and step: the slowest record
            comp = pr[:, :W, 20:, 3]
            out = {"config": cfg, "W": W, "presleep": ps, "xcds": xc, "us_per_iter_with_probe": round(ms * 1e3 / N, 3),
                   "workgroups": int(n_lat), "records_per_step": int(ok[:, 30, :].sum()),
                   "frac_same_xcd": round(float(same.sum()) / max(1, int(ok.sum())), 3),
                   "lat_ns_all": pct(d[ok]), "lat_ns_same_xcd": pct(d[same]), "lat_ns_cross_xcd": pct(d[cross]),
                   "lat_ns_slowest_record_per_wg_step": pct(wg_max[wg_max >= 0]),
                   "lat_ns_slowest_of_all_wgs_per_step": pct(wg_max.max(axis=0)),
                   "compute_cycles": pct(comp.reshape(-1))}
            print(json.dumps(out), flush=True)
            res.append(out)
os.makedirs("gpurun_out", exist_ok=True)
json.dump(res, open("gpurun_out/wg_hop.json", "w"), indent=1)
