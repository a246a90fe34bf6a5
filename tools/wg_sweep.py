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
                                   OPT_POLL_GAP, OPT_PW_ROLES, RUN_PATHS)

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
        raw = r.read_probe()
        W = dict(opts).get(OPT_WG_WAVES, 4)
    finally:
        r.close()
    WW = 1 if (W == 1 and dict(opts).get(OPT_PW_ROLES, 0) == 0) else W + 1  # waves per workgroup
    n_wg = raw.size // (N * (WW * 8 + 128))
    p4 = raw[:n_wg * WW * N * 8].reshape(n_wg, WW, N, 8).astype(np.int64)[:, :, 20:, :]
    p = p4.reshape(-1, N - 20, 8)
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
    # per workgroup: the slowest compute wave and the shortest wait -- the workgroups with the least slack set the pace
    cw = p4[:, :min(W, WW), :, :]
    wg_compute = cw[:, :, :, 3].mean(axis=2).max(axis=1)
    wg_wait = cw[:, :, :, 2].mean(axis=2).min(axis=1)
    order = np.argsort(wg_wait)
    res["least_slack_workgroups"] = [{"wg": int(i), "wait": round(float(wg_wait[i]), 0), "compute_slowest_wave": round(float(wg_compute[i]), 0),
                                      "compute_per_wave": [round(float(v), 0) for v in cw[i, :, :, 3].mean(axis=1)]} for i in order[:6]]
    res["wait_percentiles_over_workgroups"] = {q_: round(float(np.percentile(wg_wait, q_)), 0) for q_ in (0, 5, 25, 50, 75, 95, 100)}
    res["compute_percentiles_over_workgroups"] = {q_: round(float(np.percentile(wg_compute, q_)), 0) for q_ in (0, 5, 25, 50, 75, 95, 100)}
    if W == 1 and WW == 2:  # hardware placement: HW_ID of the compute wave (word 0) and of the communication wave (word 3)
        hc, hm = raw[:n_wg * 2 * N * 8].reshape(n_wg, 2, N, 8)[:, 0, 30, 0], raw[:n_wg * 2 * N * 8].reshape(n_wg, 2, N, 8)[:, 1, 30, 3]
        cu = lambda h: ((h >> 8) & 0xf) | (((h >> 12) & 1) << 4) | (((h >> 13) & 7) << 5)
        simd = lambda h: (h >> 4) & 3
        key = cu(hc.astype(np.int64)) * 4 + simd(hc.astype(np.int64))
        _, counts = np.unique(key, return_counts=True)
        res["compute_waves_per_simd_hist"] = np.bincount(counts).tolist()
        res["compute_simd_hist"] = np.bincount(simd(hc.astype(np.int64)), minlength=4).tolist()
        res["comm_simd_hist"] = np.bincount(simd(hm.astype(np.int64)), minlength=4).tolist()
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
        for W in ((4,) if quick else (2, 4)):
            for ps in ((7,) if quick else (1, 9, 17, 25, 33, 41)):
                for xc in ((0,) if (quick or g["V"] > 4000) else (1, 8)):
                    for gap in ((0,) if quick else (1, 2, 3)):
                        dual = 1
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
                o2 = [(k, v) for k, v in best[1] if k != OPT_PRESLEEP] + [(OPT_PRESLEEP, 1)]
                print("  probe", json.dumps(probe_summary(g, o2, str(o2))), flush=True)
        results.append(row)
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/wg_sweep.json", "w") as f:
        json.dump(results, f, indent=1)
    print(json.dumps([{k: v for k, v in r.items() if k != "wg"} for r in results]))


if __name__ == "__main__":
    main()
