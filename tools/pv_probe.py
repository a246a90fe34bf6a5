#!/usr/bin/env python3
"""tools/pv_probe.py -- where a step's cycles go in the patch-per-wave persistent kernel (GPU box).

For every BASELINE single-frame size: the per-step time of the lane-per-half-edge kernel and of the patch-per-wave kernel
(bit-compared), and the in-kernel probe of the latter (FLAME_NLTGV2_OPT_PROBE): per patch and step the shader cycles
spent waiting for the neighbours' records and the cycles from their arrival to the next publish.  The lock-step network
runs at the pace of its least-slack patches, so those are listed.  Output: gpurun_out/pv_probe.json."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch  # noqa: F401  (one HIP runtime per process)

import flame_amd
from flame_amd import synth
from flame_amd.regularizer import OPT_PERSISTENT, OPT_POLL_GAP, OPT_PROBE, RUN_PATHS

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


def probe_summary(g, opts):
    r = flame_amd.Regularizer(0)
    try:
        for k, v in opts:
            r.set_option(k, v)
        r.set_option(OPT_PROBE, 1)
        r.upload_graph(g)
        r.run(params, N)
        ms = r.run_timed(params, N)
        p = r.read_probe().reshape(-1, N, 8).astype(np.int64)[:, 20:, :]
        p = p[p[:, 0, 5] != 0]
    finally:
        r.close()
    wait, comp = p[:, :, 2].mean(axis=1), p[:, :, 3].mean(axis=1)
    pct = lambda a: {str(q): round(float(np.percentile(a, q)), 0) for q in (0, 5, 25, 50, 75, 95, 100)}
    order = np.argsort(wait)
    dc = np.diff(p[0, :, 5]) & 0xffffffff
    dt = np.diff(p[0, :, 6]) & 0xffffffff
    period = float(dc.mean())
    return {"patches": int(p.shape[0]), "us_per_iter_with_probe": round(ms * 1e3 / N, 3),
            "step_period_shader_cycles": round(period, 1), "step_period_us_100MHz_clock": round(float(dt.mean()) / 100.0, 4),
            "shader_clock_GHz": round(period / (float(dt.mean()) * 10.0), 3),
            "wait_cycles_over_patches": pct(wait), "compute_cycles_over_patches": pct(comp),
            "wait_fraction_of_step_mean": round(float(wait.mean()) / period, 3),
            "poll_rounds_per_step_mean": round(float(p[:, :, 4].mean()), 2),
            "least_slack_patches": [{"patch": int(i), "wait": round(float(wait[i]), 0), "compute": round(float(comp[i]), 0)} for i in order[:6]]}


def main():
    cfgs = [a for a in sys.argv[1:] if not a.startswith("--")] or ["320x240", "640x480", "1280x720"]
    results = []
    for cfg in cfgs:
        g = synth.make_graph(cfg, seed=5000)
        us_tv, path_tv, want, _ = timed(g, [(OPT_PERSISTENT, 3)])
        row = {"config": cfg, "V": int(g["V"]), "E": int(g["E"]), "vertex_per_lane_us_per_iter": round(us_tv, 3)}
        for gap in (1, 2):
            us, path, _, same = timed(g, [(OPT_PERSISTENT, 4), (OPT_POLL_GAP, gap)], want=want)
            row[f"patch_per_wave_us_per_iter_gap{gap - 1}"] = round(us, 3) if path == "persistent-pv" else path
            row["bit_identical"] = same
        us, path, _, same = timed(g, [], want=want)
        row["auto"] = {"us_per_iter": round(us, 3), "path": path, "bit_identical": same}
        row["probe"] = probe_summary(g, [(OPT_PERSISTENT, 4)])
        print(json.dumps(row), flush=True)
        results.append(row)
    # what the XCD borders cost: eight disjoint graphs, each on its own XCD (no record crosses an XCD), against one coupled
    # graph of the same size
    def small(w, h, seed):
        pos = synth.make_points(w, h, 6, seed)
        return synth.assemble_graph(pos, synth.make_data_term(pos, w, h, seed), synth.delaunay_edges_native(pos))

    g1 = synth.make_graph("640x480", seed=1234)
    g8 = synth.concat_graphs([small(228, 168, 100 + k) for k in range(8)])
    border = {"coupled_640x480": probe_summary(g1, [(OPT_PERSISTENT, 4)]),
              "eight_disjoint_228x168_one_per_xcd": probe_summary(g8, [(OPT_PERSISTENT, 4)])}
    for k, v in border.items():
        print(k, json.dumps({kk: v[kk] for kk in ("patches", "us_per_iter_with_probe", "step_period_shader_cycles")}), flush=True)
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/pv_probe.json", "w") as f:
        json.dump({"configs": results, "xcd_border_experiment": border,
                   "legend": "cycles are shader cycles (s_memtime) per step, averaged over steps 20..199 of a 200-step launch; wait = "
                             "from a patch's publish to the arrival of the last record it needs, compute = from that arrival to its "
                             "next publish; the lock-step network runs at its worst cycle mean, set by the least-slack patches"}, f, indent=1)


if __name__ == "__main__":
    main()
