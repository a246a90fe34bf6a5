import sys, os
sys.path.insert(0, os.getcwd())
import torch
import flame_amd.regularizer as R
lib = sys.argv[1]
R.library_path = lambda: os.path.join(os.getcwd(), "build", "ab", lib)
import flame_amd
from flame_amd import synth
p = flame_amd.Params()
cases = [("22x640x480", synth.concat_graphs([synth.make_graph("640x480", 100+i) for i in range(22)])),
         ("64x640x480", synth.concat_graphs([synth.make_graph("640x480", 100+i) for i in range(64)])),
         ("1920x1080", synth.make_graph("1920x1080", 1234))]
for name, g in cases:
    for tvlds in (0, 2):
        r = flame_amd.Regularizer(0); r.set_option(5, 3); r.set_option(7, tvlds); r.upload_graph(g); r.run(p, 200)
        ms = min(r.run_timed(p, 200) for _ in range(6))
        i = r.info()
        print(lib, name, "tv lds", tvlds, "us/it %.3f" % (ms*1e3/200), "groups", i["last_run_groups"], flush=True); r.close()
