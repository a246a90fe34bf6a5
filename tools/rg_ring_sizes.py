"""tools/rg_ring_sizes.py [SIZE,SIZE..] [REGIONS,REGIONS..] -- ring sizes of layout (R) per depth k (host only: no GPU).  Writes the synthetic
graph of each size to a scratch file, builds tools/rg_ring_sizes.cc against the library's own builder and prints its table."""
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from flame_amd import synth  # noqa: E402

sizes = (sys.argv[1] if len(sys.argv) > 1 else "640x480,1280x720,1920x1080").split(",")
regions = sys.argv[2] if len(sys.argv) > 2 else "256"
with tempfile.TemporaryDirectory() as tmp:
    exe = os.path.join(tmp, "rg_ring_sizes")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "flame_amd", "csrc"),
                           os.path.join(ROOT, "tools", "rg_ring_sizes.cc"), "-o", exe])
    for size in sizes:
        g = synth.make_graph(size, seed=1234)
        path = os.path.join(tmp, size + ".bin")
        with open(path, "wb") as f:
            np.array([g["V"], g["E"]], dtype=np.int32).tofile(f)
            np.ascontiguousarray(g["pos"], dtype=np.float32).tofile(f)
            np.ascontiguousarray(g["src"], dtype=np.int32).tofile(f)
            np.ascontiguousarray(g["dst"], dtype=np.int32).tofile(f)
        sys.stdout.write(subprocess.check_output([exe, path, regions]).decode().replace(tmp + "/", ""))
        sys.stdout.flush()
