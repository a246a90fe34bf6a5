"""tools/r06_ab_elide.py -- same-box A/B of the write-through elision (FLAME_NLTGV2_OPT_FAR_ELIDE = 0 / 1) on single frames, on the
uncoupled floors (eight disjoint graphs, one per XCD) and on small batches that run in the patch-per-wave form.  GPU box."""
import os
import sys
import time

import numpy as np
import torch  # noqa: F401

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import flame_amd  # noqa: E402
from flame_amd import synth  # noqa: E402
from flame_amd.regularizer import OPT_FAR_ELIDE, RUN_PATHS  # noqa: E402

P = flame_amd.Params()
N = int(os.environ.get("AB_ITERS", "2000"))


def small(w, h, seed):
    pos = synth.make_points(w, h, 6, seed)
    return synth.assemble_graph(pos, synth.make_data_term(pos, w, h, seed), synth.delaunay_edges_native(pos))


cases = [("320x240", synth.make_graph("320x240", seed=7)), ("640x480 s7", synth.make_graph("640x480", seed=7)),
         ("640x480 s1234", synth.make_graph("640x480", seed=1234)), ("1280x720 s7", synth.make_graph("1280x720", seed=7)),
         ("8 x 228x168 uncoupled", synth.concat_graphs([small(228, 168, 100 + k) for k in range(8)])),
         ("8 x 334x250 uncoupled", synth.concat_graphs([small(334, 250, 100 + k) for k in range(8)])),
         ("2 frames 640x480", synth.concat_graphs([synth.make_graph("640x480", seed=40 + k) for k in range(2)])),
         ("3 frames 640x480", synth.concat_graphs([synth.make_graph("640x480", seed=40 + k) for k in range(3)]))]
for name, g in cases:
    ref = None
    res = {0: [], 1: []}
    for rnd in range(2):
        for el in (0, 1):
            with flame_amd.Regularizer(0) as reg:
                reg.set_option(OPT_FAR_ELIDE, el)
                reg.upload_graph(g)
                reg.run(P, 200)
                out = reg.download_state(("x", "q1"))
                ref = ref or out
                same = all(np.array_equal(out[k], ref[k]) for k in ref)
                ts = []
                for _ in range(10):
                    t0 = time.perf_counter()
                    reg.run(P, N)
                    ts.append((time.perf_counter() - t0) / N * 1e6)
                info = reg.info()
                res[el].append(float(np.median(ts)))
                path, elided, per_cu = RUN_PATHS[info["last_run_path"]], info["last_run_far_elided"], info["last_run_waves_per_cu"]
            assert same, name
    a, b = min(res[0]), min(res[1])
    print(f"{name:24s} V={g['V']:6d} both copies {a:.4f} us/iter ({res[0][0]:.4f} {res[0][1]:.4f}) | elided {b:.4f} ({res[1][0]:.4f} {res[1][1]:.4f}) | "
          f"{(b / a - 1) * 100:+.1f} %  path {path} elided={elided} waves/CU {per_cu}", flush=True)
