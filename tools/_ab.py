import sys, os
sys.path.insert(0, os.getcwd())
import torch
import flame_amd.regularizer as R
lib = sys.argv[1]
R.library_path = lambda: os.path.join(os.getcwd(), "build", "ab", lib)
import flame_amd
from flame_amd import synth
p = flame_amd.Params()
for cfg, nf in (("320x240", 1), ("640x480", 1), ("1280x720", 1), ("1920x1080", 1), ("640x480", 7)):
    g = synth.make_graph(cfg, 1234) if nf == 1 else synth.concat_graphs([synth.make_graph(cfg, 100+i) for i in range(nf)])
    r = flame_amd.Regularizer(0); r.upload_graph(g); r.run(p, 200)
    ms = min(r.run_timed(p, 200) for _ in range(8))
    i = r.info()
    print(lib, cfg, nf, "us/it %.3f" % (ms*1e3/200), "path", i["last_run_path"], "he_waves", i["he_waves"], flush=True); r.close()
