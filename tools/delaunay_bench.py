#!/usr/bin/env python3
"""tools/delaunay_bench.py -- the host triangulator (flame_delaunay_triangulate) at the BASELINE sizes on this machine's cores:
the certified parallel strips against one sequential triangulation (FLAME_DELAUNAY_STRIPS=1), by thread count."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CODE = """
import sys, time, json
sys.path.insert(0, %r)
import numpy as np
from flame_amd import synth
from flame_amd.regularizer import delaunay
out = {}
for cfg, cell in (("640x480", 6), ("1280x720", 7), ("1920x1080", 6)):
    w, h = [int(v) for v in cfg.split("x")]
    pos = synth.make_points(w, h, cell, 11)
    delaunay(pos)
    ts = []
    for _ in range(15):
        t0 = time.perf_counter(); delaunay(pos); ts.append(time.perf_counter() - t0)
    out[cfg] = {"points": int(len(pos)), "ms_min": round(min(ts) * 1e3, 3), "ms_median": round(sorted(ts)[len(ts) // 2] * 1e3, 3)}
print(json.dumps(out))
""" % ROOT
rows = []
for strips, threads in ((1, 1), (0, 1), (0, 2), (0, 4), (0, 8), (0, 16), (0, 32)):
    env = dict(os.environ, FLAME_DELAUNAY_THREADS=str(threads))
    if strips:
        env["FLAME_DELAUNAY_STRIPS"] = str(strips)
    r = json.loads(subprocess.check_output([sys.executable, "-c", CODE], env=env).decode().strip().splitlines()[-1])
    rows.append({"mode": "sequential" if strips else "strips", "threads": threads, **r})
    print(json.dumps(rows[-1]), flush=True)
print(json.dumps({"host_cores": os.cpu_count()}))
