"""tools/rg_try.py SIZE[,SIZE..] [K_LIST] [REGIONS_LIST] -- the region-per-workgroup form (k_persistent_rg) on the GPU box: per size,
ring depth k and region count: bit-identity of all state arrays against the CPU checker after RG_CHECK_ITERS steps (default 57: not a
multiple of any k), microseconds per iteration (mean / min of 10 launches of 200 iterations by HIP events), and the in-kernel cycle
account per block (wait = refresh of the ring, compute = the k steps inside the workgroup).  First line per size: the planner's own choice."""
import os
import sys

import numpy as np
import torch  # noqa: F401  (first: one HIP runtime per process)

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import flame_amd  # noqa: E402
from flame_amd import synth  # noqa: E402
from flame_amd.regularizer import OPT_PERSISTENT, OPT_PROBE, OPT_RG_DEPTH, OPT_RG_REGIONS, RUN_PATHS  # noqa: E402
from oracle import capi as oracle  # noqa: E402  (the checker: this is a test tool)

sizes = sys.argv[1].split(",")
ks = [int(v) for v in (sys.argv[2] if len(sys.argv) > 2 else "1,2,3,4").split(",")]
regions = [int(v) for v in (sys.argv[3] if len(sys.argv) > 3 else "0").split(",")]
N = int(os.environ.get("RG_ITERS", "200"))
NCHK = int(os.environ.get("RG_CHECK_ITERS", "57"))
params = flame_amd.Params()
KEYS = ("x", "w1", "w2", "x_bar", "w1_bar", "w2_bar", "x_prev", "w1_prev", "w2_prev", "q1", "q2", "q3")


def timed(reg):
    reg.run(params, N)
    ts = [reg.run_timed(params, N) for _ in range(10)]
    return np.mean(ts) * 1e3 / N, min(ts) * 1e3 / N


for size in sizes:
    g = synth.make_graph(size, seed=int(os.environ.get("RG_SEED", "1234")))
    ref = synth.copy_graph(g)
    oracle.run(ref, NCHK)
    with flame_amd.Regularizer(0) as reg:
        reg.upload_graph(g)
        us, mn = timed(reg)
        info = reg.info()
        print(f"{size} V {g['V']} E {g['E']}: planner -> {RUN_PATHS[info['last_run_path']]} {us:.4f} us/iter (min {mn:.4f})", flush=True)
    for nr in regions:
        for k in ks:
            with flame_amd.Regularizer(0) as reg:
                reg.set_option(OPT_PERSISTENT, 7)
                reg.set_option(OPT_RG_DEPTH, k)
                reg.set_option(OPT_RG_REGIONS, nr)
                reg.upload_graph(g)
                reg.run(params, NCHK)
                out = reg.download_state()
                info = reg.info()
                path = RUN_PATHS[info["last_run_path"]]
                bad = [key for key in KEYS if not np.array_equal(out[key], ref[key])]
                if info["last_run_path"] != 8:
                    print(f"{size} k {k} regions {nr}: the form did not run (path {path}); identical {not bad}", flush=True)
                    continue
                us, mn = timed(reg)
                info = reg.info()
                acct = ""
                if os.environ.get("RG_PROBE", "1") != "0":
                    reg.set_option(OPT_PROBE, 1)
                    reg.run(params, N)
                    nb = (N + k - 1) // k
                    pa = reg.read_probe().reshape(-1, nb, 16).astype(np.int64)
                    p = pa[:, nb // 4:-1, :]  # (steady state; the last block may be short)
                    wait, comp, spins = p[:, :, 2].mean(axis=1), p[:, :, 3].mean(axis=1), p[:, :, 4].mean(axis=1)
                    ph = [np.median(p[:, :, c].mean(axis=1)) / k for c in (8, 9, 10, 11)]
                    acct = (f" | per block: wait median {np.median(wait):.0f} min {wait.min():.0f} max {wait.max():.0f} (first poll round back after {np.median(p[:, :, 0]):.0f}), "
                            f"compute median {np.median(comp):.0f} max {comp.max():.0f} min {comp.min():.0f} cycles, poll rounds {spins.mean():.1f}; per step (wave 0): "
                            f"E {ph[0]:.0f} barrier {ph[1]:.0f} V {ph[2]:.0f} barrier {ph[3]:.0f}; sc1 store ack {np.median(pa[:, 0, 12]):.0f}, lone sc1 load {np.median(pa[:, 0, 13]):.0f} cycles")
                    reg.set_option(OPT_PROBE, 0)
                print(f"{size} k {k} regions {nr or 'CUs'}: {us:.4f} us/iter (min {mn:.4f}) {path} timeouts {info['timeouts_recovered']} "
                      f"bit-identical {not bad}{'' if not bad else ' DIFFERS: ' + ','.join(bad)}{acct}", flush=True)
