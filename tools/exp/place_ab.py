"""A/B of record placement (FLAME_NLTGV2_OPT_PLACEMENT) on one box: per-iteration time with and without, same graphs."""
import os, sys, time
import numpy as np
import torch  # noqa: F401  (first: one HIP runtime per process)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import flame_amd
from flame_amd import synth
from flame_amd.regularizer import OPT_PLACEMENT

sizes = sys.argv[1:] or ["640x480"]
for size in sizes:
    for seed in (7, 8, 9):
        g = synth.make_graph(size, seed=seed)
        ref = None
        for rnd in range(2):
            for place in (1, 2):
                with flame_amd.Regularizer(0) as reg:
                    reg.set_option(OPT_PLACEMENT, min(place, 1))
                    reg.upload_graph(g)
                    t0 = time.perf_counter()
                    reg.run(flame_amd.Params(), 200)
                    first = (time.perf_counter() - t0) * 1e3
                    ts = []
                    for _ in range(12):
                        t0 = time.perf_counter()
                        reg.run(flame_amd.Params(), 2000)
                        ts.append((time.perf_counter() - t0) / 2000 * 1e6)
                    out = reg.download_state()
                    info = reg.info()
                    pi = reg.placement_info()
                    if False:
                        from flame_amd.regularizer import OPT_PROBE
                        reg.set_option(OPT_PROBE, 1)
                        reg.run(flame_amd.Params(), 20)
                        pr = reg.read_probe().reshape(-1, 20, 8)
                        n = pr.shape[0]
                        per = (n + 7) // 8
                        xcc = pr[:, 5, 1] & 15
                        print("   true XCC == patch // per_xcd for", int((xcc == np.arange(n) // per).sum()), "of", n, "patches")
                        for k in range(8):
                            sl = pr[k * per:(k + 1) * per]
                            print("   slot", k, "xcc ids", np.unique(sl[:, 5, 1], return_counts=True), "raw", hex(int(sl[0, 5, 1])), "hwid", hex(int(sl[0, 5, 0])))
                        reg.set_option(OPT_PROBE, 0)
                    bad = reg.layout_selftest()
                if ref is None:
                    ref = out
                same = all(np.array_equal(out[k], ref[k]) for k in ref)
                print(f"{size} seed {seed} place {place}: {np.median(ts):.4f} us/iter (min {min(ts):.4f}) first run {first:.1f} ms "
                      f"path {info.get('last_run_path')} same={same} selftest={bad} {pi}", flush=True)
