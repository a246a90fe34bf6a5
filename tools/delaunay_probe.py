"""tools/delaunay_probe.py -- the host triangulator by strip count and method on this host: merged strips (round 5, the default) against the
certified strips (FLAME_DELAUNAY_MERGE=0) and one thread; FLAME_DELAUNAY_PROFILE=1 prints the phases."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from flame_amd import synth  # noqa: E402
from flame_amd.regularizer import delaunay  # noqa: E402

for size in (sys.argv[1].split(",") if len(sys.argv) > 1 else ("640x480", "1280x720", "1920x1080")):
    pos = np.ascontiguousarray(synth.make_graph(size, seed=1234)["pos"], dtype=np.float32)
    for merge in ("1", "0"):
        for strips in ("", "1", "8", "16", "32", "64"):
            if strips == "1" and merge == "0":
                continue
            os.environ["FLAME_DELAUNAY_MERGE"] = merge
            if strips:
                os.environ["FLAME_DELAUNAY_STRIPS"] = strips
            else:
                os.environ.pop("FLAME_DELAUNAY_STRIPS", None)
            ts = []
            for _ in range(15):
                t = time.perf_counter()
                delaunay(pos)
                ts.append((time.perf_counter() - t) * 1e3)
            print(f"{size} {len(pos)} points, {'merged' if merge == '1' else 'certified'} strips {strips or 'default'}: median {np.median(ts):.3f} min {min(ts):.3f} ms", flush=True)
