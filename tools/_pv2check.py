import sys, os
sys.path.insert(0, "/root/repo")
import numpy as np
import torch  # noqa
import flame_amd
from flame_amd import synth
from flame_amd.regularizer import RUN_PATHS
from oracle import capi as oracle
p = flame_amd.Params()
for cfg, nf, n in (("640x480", 1, 37), ("1920x1080", 1, 41), ("640x480", 5, 30), ("640x480", 9, 22), ("1280x720", 1, 25)):
    frames = [synth.make_graph(cfg, seed=60 + i) for i in range(nf)]
    g = frames[0] if nf == 1 else synth.concat_graphs(frames)
    ref = synth.copy_graph(g); oracle.run(ref, n)
    for form in (1, 6):
        with flame_amd.Regularizer(0) as r:
            r.set_option(5, form)
            r.upload_graph(g)
            st = r.layout_selftest()
            r.run(p, n)
            out = r.download_state()
            i = r.info()
            same = all(np.array_equal(out[k], ref[k]) for k in ("x", "w1", "w2", "x_bar", "w1_bar", "w2_bar", "q1", "q2", "q3"))
            print(cfg, nf, "form", form, RUN_PATHS[i["last_run_path"]], "groups", i["last_run_groups"], "selftest", st, "bit-identical", same, "timeouts", i["timeouts_recovered"], flush=True)
