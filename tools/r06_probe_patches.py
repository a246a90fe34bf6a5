"""tools/r06_probe_patches.py -- which patches pace the lock-step network of k_persistent_pv: per patch the probe's compute and wait, its
fetch-list length and largest degree; the least-slack patches.  GPU box.  (profiles/r06_probe_patches.txt was taken with the
write-through elision of commit b4516bc beside it: the `elision 0 / 1` lines.)"""
import os
import sys

import numpy as np
import torch  # noqa: F401

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import flame_amd  # noqa: E402
from flame_amd import synth  # noqa: E402
from flame_amd.regularizer import OPT_PERSISTENT, OPT_PROBE  # noqa: E402

P = flame_amd.Params()
N = 200


def small(w, h, seed):
    pos = synth.make_points(w, h, 6, seed)
    return synth.assemble_graph(pos, synth.make_data_term(pos, w, h, seed), synth.delaunay_edges_native(pos))


cases = [("640x480 s1234 (coupled)", synth.make_graph("640x480", seed=1234)),
         ("8 x 228x168 (uncoupled)", synth.concat_graphs([small(228, 168, 100 + k) for k in range(8)]))]
for name, g in cases:
    for el in (0,):
        with flame_amd.Regularizer(0) as reg:
            reg.set_option(OPT_PERSISTENT, 4)
            reg.upload_graph(g)
            reg.run(P, N)
            plain = min(reg.run_timed(P, 2000) for _ in range(5)) * 1e3 / 2000
            reg.set_option(OPT_PROBE, 1)
            reg.run(P, N)
            ms = reg.run_timed(P, N)
            p = reg.read_probe().reshape(-1, N, 8).astype(np.int64)[:, 20:, :]
            p = p[p[:, 0, 5] != 0]
        wait, comp = p[:, :, 2].mean(axis=1), p[:, :, 3].mean(axis=1)
        nf, deg, xcc = (p[:, 0, 7] >> 8) & 0xff, p[:, 0, 7] & 0xff, p[:, 0, 1]
        period = float((np.diff(p[0, :, 5]) & 0xffffffff).mean())
        rounds = p[:, :, 4].mean(axis=1)
        print(f"== {name}: {plain:.4f} us/iter unprobed; probed {ms * 1e3 / N:.4f} us = {period:.0f} cycles; {p.shape[0]} patches, "
              f"fetch list mean {nf.mean():.1f} max {nf.max()}, compute median {np.median(comp):.0f} max {comp.max():.0f}, wait median {np.median(wait):.0f} min {wait.min():.0f}")
        order = np.argsort(wait)[:10]
        print("   least-slack patches (wait, compute, sum, fetch list, largest degree, xcc, poll rounds): " +
              "; ".join(f"{wait[i]:.0f}+{comp[i]:.0f}={wait[i] + comp[i]:.0f} f{nf[i]} d{deg[i]} x{xcc[i]} r{rounds[i]:.1f}" for i in order))
        for lo, hi in ((0, 8), (8, 16), (16, 24), (24, 32), (32, 65)):
            m = (nf >= lo) & (nf < hi)
            if m.any():
                print(f"   fetch list {lo:2d}..{hi - 1:2d}: {int(m.sum()):4d} patches, wait mean {wait[m].mean():.0f} min {wait[m].min():.0f}, compute mean {comp[m].mean():.0f}")
        c = np.corrcoef(np.stack([wait, comp, nf.astype(float), deg.astype(float)]))
        print(f"   correlation of the wait with compute {c[0, 1]:+.2f}, with the fetch-list length {c[0, 2]:+.2f}, with the largest degree {c[0, 3]:+.2f}")
