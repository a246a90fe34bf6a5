"""tools/uncoupled.py -- period of 8 DISJOINT graphs (one per XCD: no record crosses an XCD) against coupled graphs of the same total size."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch  # noqa
import flame_amd
from flame_amd import synth
from flame_amd.regularizer import OPT_PERSISTENT, RUN_PATHS
params = flame_amd.Params()
N = 200
def small(w, h, seed):
    pos = synth.make_points(w, h, 6, seed)
    return synth.assemble_graph(pos, synth.make_data_term(pos, w, h, seed), synth.delaunay_edges_native(pos))
def timeit(g):
    r = flame_amd.Regularizer(0)
    r.set_option(OPT_PERSISTENT, 4)
    r.upload_graph(g); r.run(params, N); r.run(params, N)
    ts = [r.run_timed(params, N) for _ in range(10)]
    i = r.info(); r.close()
    return np.mean(ts) * 1e3 / N, RUN_PATHS[i["last_run_path"]], i["patches"]
for label, (w, h) in (("x1", (228, 168)), ("x2", (320, 240)), ("720p", (334, 250)), ("x4", (452, 338)), ("x5", (506, 378)), ("x6", (554, 414))):
    g8 = synth.concat_graphs([small(w, h, 100 + k) for k in range(8)])
    t, path, patches = timeit(g8)
    print(f"{label:5s} uncoupled: V={g8['V']} {t:.4f} us/iter {path} patches {patches}", flush=True)
