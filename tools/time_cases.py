"""tools/time_cases.py CASE.. -- us per iteration of the automatic path (mean / min of 10 launches of 200 iterations), with a state hash."""
import hashlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch  # noqa
import flame_amd
from flame_amd import synth
from flame_amd.regularizer import RUN_PATHS
params = flame_amd.Params()
N = int(os.environ.get("TC_ITERS", "200"))
opts = [tuple(int(t) for t in kv.split("=")) for kv in os.environ.get("TC_OPTS", "").split(",") if kv]
for c in sys.argv[1:]:
    cfg, nf = c.split(":"); nf = int(nf)
    frames = [synth.make_graph(cfg, seed=1234 + i) for i in range(nf)]
    g = frames[0] if nf == 1 else synth.concat_graphs(frames)
    r = flame_amd.Regularizer(0)
    for k, v in opts: r.set_option(k, v)
    r.upload_graph(g)
    r.run(params, N)
    st = r.download_state(("x", "w1", "q1"))
    h = hashlib.sha1(b"".join(st[k].tobytes() for k in ("x", "w1", "q1"))).hexdigest()[:10]
    r.run(params, N)
    ts = [r.run_timed(params, N) for _ in range(10)]
    info = r.info()
    print(f"{c:14s} {np.mean(ts) * 1e3 / N:.4f} us/iter (min {min(ts) * 1e3 / N:.4f}) {RUN_PATHS[info['last_run_path']]} patches {info['patches']} timeouts {info['timeouts_recovered']} hash {h}", flush=True)
    r.close()
