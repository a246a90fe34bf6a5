"""tools/ab.py SIZE[,SIZE..] OPT=V[,OPT=V..] [OPT=V..] ... -- same-box A/B of option settings of the solver (run on the GPU box).
Each argument after the sizes is one setting (comma-separated option=value pairs by number, e.g. 17=0 or 16=0,17=0; "-" =
defaults).  Per size and graph: microseconds per iteration (median of 12 runs of AB_ITERS = 2000 iterations, host clock around the call), bit-identity to the first
setting, and the in-kernel cycle account (median compute / least-slack wait) where the patch-per-wave form ran."""
import os
import sys
import time

import numpy as np
import torch  # noqa: F401  (first: one HIP runtime per process)

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import flame_amd  # noqa: E402
from flame_amd import synth  # noqa: E402
from flame_amd.regularizer import OPT_PROBE  # noqa: E402

sizes = sys.argv[1].split(",")
settings = [[] if a == "-" else [tuple(int(x) for x in kv.split("=")) for kv in a.split(",")] for a in (sys.argv[2:] or ["-"])]
for size in sizes:
    for seed in (7, 8):
        g = synth.make_graph(size, seed=seed)
        ref = None
        for rnd in range(2):
            for st in settings:
                with flame_amd.Regularizer(0) as reg:
                    for k, v in st:
                        reg.set_option(k, v)
                    reg.upload_graph(g)
                    reg.run(flame_amd.Params(), 200)
                    out = reg.download_state()
                    if ref is None:
                        ref = out
                    same = all(np.array_equal(out[k], ref[k]) for k in ref)
                    ts = []
                    n_it = int(os.environ.get("AB_ITERS", "2000"))  # iterations per timed run (one launch each)
                    for _ in range(12):
                        t0 = time.perf_counter()
                        reg.run(flame_amd.Params(), n_it)
                        ts.append((time.perf_counter() - t0) / n_it * 1e6)
                    info = reg.info()
                    acct = ""
                    if rnd == 0 and info["last_run_path"] == 6:
                        reg.set_option(OPT_PROBE, 1)
                        reg.run(flame_amd.Params(), 200)
                        p = reg.read_probe().reshape(-1, 200, 8).astype(np.int64)[:, 20:, :]
                        p = p[p[:, 0, 5] != 0]
                        acct = f" | probe: compute median {np.median(p[:, :, 3].mean(axis=1)):.0f} max {p[:, :, 3].mean(axis=1).max():.0f}, wait min {p[:, :, 2].mean(axis=1).min():.0f}"
                print(f"{size} seed {seed} {st or 'defaults'}: {np.median(ts):.4f} us/iter (min {min(ts):.4f}) path {info['last_run_path']} "
                      f"patches {info['patches']} same={same}{acct}", flush=True)
