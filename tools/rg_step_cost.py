"""tools/rg_step_cost.py -- what ONE step costs inside a workgroup of k_persistent_rg, without any exchange: tiny graphs run as a single
region (nothing to fetch, nothing to publish), 20 000 steps per launch; the time per step is the E phase + the V phase + two barriers.
Sizes: W x H images of 6 px cells (W/6 x H/6 vertices)."""
import os
import sys

import numpy as np
import torch  # noqa: F401

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import flame_amd  # noqa: E402
from flame_amd import synth  # noqa: E402
from flame_amd.regularizer import OPT_PERSISTENT, OPT_PROBE, OPT_RG_DEPTH, OPT_RG_REGIONS, RUN_PATHS  # noqa: E402
from oracle import capi as oracle  # noqa: E402

N = int(os.environ.get("RG_ITERS", "20000"))
params = flame_amd.Params()
for spec in (sys.argv[1:] or ["48x42", "66x54", "96x66", "120x96"]):
    w, h = (int(v) for v in spec.split("x"))
    synth.CONFIGS[spec] = (w, h, 6)
    g = synth.make_graph(spec, seed=5)
    ref = synth.copy_graph(g)
    oracle.run(ref, 100)
    for depth in (1, 6):
      with flame_amd.Regularizer(0) as reg:
        reg.set_option(OPT_PERSISTENT, 7)
        reg.set_option(OPT_RG_DEPTH, depth)
        reg.set_option(OPT_RG_REGIONS, 1)
        reg.upload_graph(g)
        reg.run(params, 100)
        out = reg.download_state()
        same = all(np.array_equal(out[k], ref[k]) for k in ("x", "w1", "w2", "q1", "q2", "q3"))
        path = RUN_PATHS[reg.info()["last_run_path"]]
        reg.run(params, N)
        ts = [reg.run_timed(params, N) for _ in range(5)]
        us = min(ts) * 1e3 / N
        reg.set_option(OPT_PROBE, 1)
        reg.run(params, 600)
        nb = (600 + depth - 1) // depth
        pr = reg.read_probe().reshape(-1, nb, 16).astype(np.int64)[:, nb // 4:-1, :]
        acct = (f"probe per block: wait {pr[:, :, 2].mean():.0f} (first round {pr[:, :, 0].mean():.0f}) compute {pr[:, :, 3].mean():.0f}; per step E {pr[:, :, 8].mean() / depth:.0f} "
                f"barrier {pr[:, :, 9].mean() / depth:.0f} V {pr[:, :, 10].mean() / depth:.0f} barrier {pr[:, :, 11].mean() / depth:.0f} cycles; "
                f"cycles per 100 MHz tick {(pr[0, -1, 5] - pr[0, 0, 5]) % (1 << 32) / max(1, (pr[0, -1, 6] - pr[0, 0, 6]) % (1 << 32)):.2f}")
        reg.set_option(OPT_PROBE, 0)
        deg = np.bincount(np.concatenate([g["src"], g["dst"]]), minlength=g["V"]).max()
        print(f"{spec} blocks of {depth}: V {g['V']} ({(g['V'] + 63) // 64} V-waves) E {g['E']} ({(g['E'] + 63) // 64} E-waves) max degree {deg}: {us * 1e3:.1f} ns per step "
              f"= {us * 2400:.0f} cycles at 2.4 GHz; {path}; bit-identical {same} | {acct}", flush=True)
