#!/usr/bin/env python3
"""tools/wg_sweep.py -- A/B of the persistent forms on the single-frame configs (GPU box): lane-per-half-edge
(k_persistent_he) against patch-per-workgroup (k_persistent_wg) over waves per workgroup, pre-poll sleep and XCD
count; every variant's result is compared bit for bit with the lane-per-half-edge result.  With --probe, the
in-kernel cycle probe of the patch-per-workgroup kernel is summarised (where a step's cycles go)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch  # noqa: F401  (one HIP runtime per process)

import flame_amd
from flame_amd import synth
from flame_amd.regularizer import (OPT_PERSISTENT, OPT_PRESLEEP, OPT_XCDS, OPT_WG_WAVES, OPT_PROBE, OPT_DUAL_PUBLISH,
                                   OPT_POLL_GAP, RUN_PATHS)

params = flame_amd.Params()
N = 200


def timed(g, opts, reps=5, want=None):
    r = flame_amd.Regularizer(0)
    try:
        for k, v in opts:
            r.set_option(k, v)
        r.upload_graph(g)
        r.run(params, N)
        out = r.download_state(("x", "q1"))
        r.upload_graph(g)
        r.run(params, N)
        ms = min(r.run_timed(params, N) for _ in range(reps))
        info = r.info()
        same = None if want is None else bool(np.array_equal(out["x"], want["x"]) and np.array_equal(out["q1"], want["q1"]))
        return ms * 1e3 / N, RUN_PATHS[info["last_run_path"]], out, same
    finally:
        r.close()


def probe_summary(g, opts, label):
    r = flame_amd.Regularizer(0)
    try:
        for k, v in opts:
            r.set_option(k, v)
        r.set_option(OPT_PROBE, 1)
        r.upload_graph(g)
        r.run(params, N)
        r.run(params, N)
        p = r.read_probe().reshape(-1, N, 8).astype(np.int64)
    finally:
        r.close()
    p = p[:, 20:, :]
    comm = p[:, 0, 7] == 1
    res = {"label": label, "waves": int(p.shape[0]), "comm_waves": int(comm.sum())}
    q = p[comm]
    if q.size:
        res["comm_wave_cycles"] = {"presleep": round(float(q[:, :, 0].mean()), 1), "poll": round(float(q[:, :, 1].mean()), 1),
                                   "barrier_wait": round(float(q[:, :, 2].mean()), 1),
                                   "poll_rounds_per_step": round(float(q[:, :, 4].mean()), 2)}
    q = p[~comm]
    if q.size:
        res["compute_wave_cycles"] = {"barrier_wait": round(float(q[:, :, 2].mean()), 1), "compute": round(float(q[:, :, 3].mean()), 1),
                                      "compute_p10": round(float(np.percentile(q[:, :, 3], 10)), 1),
                                      "compute_p90": round(float(np.percentile(q[:, :, 3], 90)), 1)}
    t = p[0, :, 6]
    dt = np.diff(t) & 0xffffffff
    res["step_period_us_100MHz_clock"] = round(float(dt.mean()) / 100.0, 4)
    c = p[0, :, 5]
    dc = np.diff(c) & 0xffffffff
    res["step_period_shader_cycles"] = round(float(dc.mean()), 1)
    return res


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    do_probe = "--probe" in sys.argv
    quick = "--quick" in sys.argv
    cfgs = args or ["640x480", "1280x720", "320x240", "1920x1080"]
    results = []
    for cfg in cfgs:
        g = synth.make_graph(cfg, seed=5000)
        us_he, path, want, _ = timed(g, [(OPT_PERSISTENT, 2)])
        row = {"config": cfg, "V": int(g["V"]), "E": int(g["E"]), "he_us_per_iter": round(us_he, 3), "wg": []}
        print(cfg, "V", g["V"], "he", f"{us_he:.3f} us/it", path, flush=True)
        best = None
        for W in ((4,) if quick else (1, 2, 4, 8)):
            for ps in ((7,) if quick else (1, 5, 9, 13)):
                for xc in ((0,) if (quick or g["V"] > 4000) else (0, 1, 8)):
                    for gap in ((0,) if quick else (1, 2, 3, 5)):
                        dual = 1
                        if xc != 0 and (gap != 2 or ps != 5):
                            continue
                        opts = [(OPT_PERSISTENT, 4), (OPT_WG_WAVES, W), (OPT_PRESLEEP, ps), (OPT_XCDS, xc), (OPT_POLL_GAP, gap)]
                        try:
                            us, path, _, same = timed(g, opts, want=want)
                        except Exception as e:  # noqa: BLE001
                            print("  W", W, "ps", ps, "xcds", xc, "gap", gap, "ERR", e, flush=True)
                            continue
                        rec = {"W": W, "presleep": ps, "xcds": xc, "gap": gap, "us_per_iter": round(us, 3), "path": path, "bit_identical": same}
                        row["wg"].append(rec)
                        print("  ", json.dumps(rec), flush=True)
                        if path == "persistent-wg" and same and (best is None or us < best[0]):
                            best = (us, opts)
        if best:
            row["best_wg_us_per_iter"] = round(best[0], 3)
            row["speedup_vs_he"] = round(us_he / best[0], 3)
            if do_probe:
                row["probe"] = probe_summary(g, best[1], str(best[1]))
                print("  probe", json.dumps(row["probe"]), flush=True)
        results.append(row)
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/wg_sweep.json", "w") as f:
        json.dump(results, f, indent=1)
    print(json.dumps([{k: v for k, v in r.items() if k != "wg"} for r in results]))


if __name__ == "__main__":
    main()
