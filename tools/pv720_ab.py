import os, sys
sys.path.insert(0, "/root/repo")
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
