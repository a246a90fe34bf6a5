#!/usr/bin/env python3
"""tools/pv_big.py -- the patch-per-wave persistent kernel on graphs beyond 12 patches per CU (GPU box).

For each (config, frames) case: the automatic path, the vertex-per-lane kernel and the patch-per-wave kernels (one / two half-edges per lane), timed
(mean of the launches after the first) and bit-compared.  PV_VARIANTS="113=4,108=9;..." adds columns with option settings."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch  # noqa: F401  (one HIP runtime per process)

import flame_amd
from flame_amd import synth
from flame_amd.regularizer import OPT_PERSISTENT, RUN_PATHS

params = flame_amd.Params()
N = 200


def timed(g, opts, reps=6, want=None):
    r = flame_amd.Regularizer(0)
    try:
        for k, v in opts:
            r.set_option(k, v)
        r.upload_graph(g)
        r.run(params, N)
        out = r.download_state(("x", "w1", "w2", "q1", "q2", "q3"))
        info0 = r.info()
        r.upload_graph(g)
        r.run(params, N)
        ts = [r.run_timed(params, N) for _ in range(reps)]
        info = r.info()
        same = None if want is None else bool(all(np.array_equal(out[k], want[k]) for k in out))
        return {"us_per_iter_mean": round(float(np.mean(ts)) * 1e3 / N, 3), "us_per_iter_min": round(min(ts) * 1e3 / N, 3),
                "path": RUN_PATHS[info["last_run_path"]], "first_path": RUN_PATHS[info0["last_run_path"]], "patches": info["patches"],
                "groups": info["last_run_groups"], "timeouts_recovered": info["timeouts_recovered"], "bit_identical": same}, out
    finally:
        r.close()


def main():
    cases = [a for a in sys.argv[1:]] or ["1920x1080:1", "1280x720:1", "640x480:3", "640x480:7"]
    for c in cases:
        cfg, nf = c.split(":")
        nf = int(nf)
        frames = [synth.make_graph(cfg, seed=1234 + i) for i in range(nf)]
        g = frames[0] if nf == 1 else synth.concat_graphs(frames)
        row = {"case": c, "V": int(g["V"]), "E": int(g["E"])}
        tv, want = timed(g, [(OPT_PERSISTENT, 3)])  # the vertex-per-lane kernel: the other persistent form, and the bit reference
        row["tv"] = tv
        pv, _ = timed(g, [(OPT_PERSISTENT, 4)], want=want)
        row["pv"] = pv
        pv2, _ = timed(g, [(OPT_PERSISTENT, 6)], want=want)  # two half-edges per lane
        row["pv2"] = pv2
        for spec in [s for s in os.environ.get("PV_VARIANTS", "").split(";") if s]:  # e.g. "13=4,8=9;13=4,8=17"
            opts = [(OPT_PERSISTENT, 4)] + [tuple(int(t) for t in kv.split("=")) for kv in spec.split(",")]
            row["pv[" + spec + "]"], _ = timed(g, opts, want=want)
        auto, _ = timed(g, [], want=want)
        row["auto"] = auto
        B = 64 * g["V"] + 40 * g["E"]
        row["auto_frac_of_8TBps"] = round(B / (auto["us_per_iter_mean"] * 1e-6) / 8e12, 3)
        print(json.dumps(row), flush=True)


if __name__ == "__main__":
    main()
