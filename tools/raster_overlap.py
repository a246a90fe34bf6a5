"""How interpolate_mesh_begin/_end overlap with a solver chunk: wall time from begin to the return of _end, for several chunk lengths."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
import torch
import flame_amd
from flame_amd import synth

size = sys.argv[1] if len(sys.argv) > 1 else "1920x1080"
W, H, _ = synth.CONFIGS[size]
g = synth.make_graph(size, seed=3)
tris, edges = flame_amd.delaunay(g["pos"])
P = flame_amd.Params()
reg = flame_amd.Regularizer(0)
stream = torch.cuda.Stream(priority=-1)
reg.set_stream(stream.cuda_stream)
reg.upload_graph(g)
reg.run(P, 200)
for n in (0, 200, 400, 800, 1600, 3200):
    rows = []
    for rep in range(6):
        reg.sync()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        reg.interpolate_mesh_begin(tris, H, W)
        t1 = time.perf_counter()
        e0.record(stream)
        if n:
            reg.run_async(P, n)
        e1.record(stream)
        t2 = time.perf_counter()
        reg.interpolate_mesh_end(copy=False)
        t3 = time.perf_counter()
        torch.cuda.synchronize()
        rows.append(((t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t0) * 1e3, e0.elapsed_time(e1)))
    r = np.median(np.array(rows), axis=0)
    print("%s chunk %5d it: begin call %.3f ms, launch %.3f ms, begin -> end returned %.3f ms, solver chunk %.3f ms" % (size, n, r[0], r[1], r[2], r[3]))
