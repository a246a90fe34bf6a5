import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
from flame_amd import synth
from flame_amd.regularizer import delaunay
pos = np.ascontiguousarray(synth.make_graph(sys.argv[1], seed=1234)["pos"], dtype=np.float32)
buf = (np.empty((2 * len(pos), 3), np.int32), np.empty((3 * len(pos), 2), np.int32))
for strips in ("", "16", "64"):
    if strips: os.environ["FLAME_DELAUNAY_STRIPS"] = strips
    else: os.environ.pop("FLAME_DELAUNAY_STRIPS", None)
    ts = []
    for _ in range(101):
        t = time.perf_counter(); delaunay(pos, out=buf); ts.append((time.perf_counter() - t) * 1e3)
    ts = sorted(ts[1:])
    tp = []
    for _ in range(60):
        time.sleep(0.002); t = time.perf_counter(); delaunay(pos, out=buf); tp.append((time.perf_counter() - t) * 1e3)
    tp.sort()
    print(f"threads {os.environ.get('FLAME_DELAUNAY_THREADS','default')} strips {strips or 'default'}: back to back median {ts[50]:.3f} p95 {ts[94]:.3f} | 2 ms apart median {tp[30]:.3f} p95 {tp[56]:.3f}", flush=True)
