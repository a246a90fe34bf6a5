"""tools/context_order_ab.py -- three contexts per size, one after the other in one process: the second and third take the first one's record-placement
pool (and its page ranking) over; with FLAME_NLTGV2_LAZY_CALIBRATION=1 the ranking is measured in the first run instead of in create()."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import flame_amd
from flame_amd import synth
from flame_amd.regularizer import RUN_PATHS
p = flame_amd.Params()
for size in ("1280x720", "640x480"):
    g = synth.make_graph(size, seed=1234)
    for rep in range(3):
        with flame_amd.Regularizer(0) as reg:
            reg.upload_graph(g)
            reg.run(p, 200)
            ts = [reg.run_timed(p, 200) for _ in range(10)]
            print(size, os.environ.get("FLAME_NLTGV2_LAZY_CALIBRATION", "-"), f"{np.mean(ts)*5:.4f} us/iter", RUN_PATHS[reg.info()["last_run_path"]], reg.placement_info(), flush=True)
