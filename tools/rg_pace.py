"""tools/rg_pace.py SIZE K -- poll pacing of k_persistent_rg: us per iteration by (pre-sleep, gap) in units of 64 cycles."""
import os, sys
import numpy as np
import torch  # noqa: F401
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import flame_amd
from flame_amd import synth
from flame_amd.regularizer import OPT_PERSISTENT, OPT_RG_DEPTH, OPT_PRESLEEP, OPT_POLL_GAP
size, k = sys.argv[1], int(sys.argv[2])
g = synth.make_graph(size, seed=1234)
params = flame_amd.Params()
for pre in (0, 4, 8, 12, 16, 20, 24):
    row = []
    for gap in (0, 1, 2, 4):
        with flame_amd.Regularizer(0) as reg:
            reg.set_option(OPT_PERSISTENT, 7); reg.set_option(OPT_RG_DEPTH, k)
            reg.set_option(OPT_PRESLEEP, pre + 1); reg.set_option(OPT_POLL_GAP, gap + 1)
            reg.upload_graph(g)
            reg.run(params, 200)
            ts = [reg.run_timed(params, 200) for _ in range(8)]
            row.append(f"gap {gap}: {np.mean(ts) * 5:.4f}")
    print(f"{size} k {k} pre-sleep {pre}: " + "  ".join(row), flush=True)
