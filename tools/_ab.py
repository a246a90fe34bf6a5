import sys, os
sys.path.insert(0, os.getcwd())
import torch
import flame_amd.regularizer as R
lib = sys.argv[1]
R.library_path = lambda: os.path.join(os.getcwd(), "build", "ab", lib)
import flame_amd
from flame_amd import synth
p = flame_amd.Params()
gs = [synth.make_graph("640x480", 100+i) for i in range(64)]
for nf in (16, 22, 30, 44, 64):
    g = synth.concat_graphs(gs[:nf])
    r = flame_amd.Regularizer(0); r.upload_graph(g); r.run(p, 200)
    ms = min(r.run_timed(p, 200) for _ in range(6))
    i = r.info()
    print(lib, nf, "frames: us/it %.3f" % (ms*1e3/200), "path", i["last_run_path"], "groups", i["last_run_groups"], "frame-iters/s %.2fM" % (nf/(ms*1e-3/200)/1e6), "frac %.3f" % (i["algorithmic_bytes_per_iter"]/(ms*1e-3/200)/8e12), flush=True); r.close()
