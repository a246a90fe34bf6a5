"""tools/patch_lead.py [SIZE...] -- how far the patches of the patch-per-wave kernel are apart while it runs: the probe stamps the end of every step
of every patch with the 100 MHz wall clock; for every step the spread of its end over the patches, in periods, is how many iterations the first
patch is AHEAD of the last one at that moment -- the number an open run's margin (kOpenMargin, nltgv2_persistent.hip) has to exceed together with the
interval of its checks.  GPU box."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch  # noqa: F401

import flame_amd
from flame_amd import synth
from flame_amd.regularizer import OPT_PERSISTENT, OPT_PROBE

P = flame_amd.Params()
N = 400
for size in (sys.argv[1:] or ["640x480", "1280x720", "320x240"]):
    g = synth.make_graph(size, seed=1)
    with flame_amd.Regularizer(0) as reg:
        reg.set_option(OPT_PERSISTENT, 4)
        reg.upload_graph(g)
        reg.run(P, N)
        reg.set_option(OPT_PROBE, 1)
        reg.run(P, N)
        words = reg.read_probe()
        n_patches = len(words) // (N * 8)
        pr = words[: n_patches * N * 8].reshape(n_patches, N, 8)
        live = pr[:, :, 6].max(axis=1) != 0
        t = pr[live][:, :, 6].astype(np.int64)                  # [patch][step] end of the step, 10 ns units (32 bits: unwrap)
        t = t - t[:, :1].min()
        t = np.where(t < 0, t + (1 << 32), t)
        period = float(np.median(np.diff(t, axis=1)))            # 10 ns units per iteration
        spread = (t.max(axis=0) - t.min(axis=0)) / period        # per step: first patch's lead over the last, in iterations
        steady = spread[20:]
        print("%s: %d patches, period %.3f us; lead of the first patch over the last (iterations): steps 0-19 max %.1f, afterwards median %.2f, max %.2f"
              % (size, int(live.sum()), period / 100.0, spread[:20].max(), float(np.median(steady)), float(steady.max())))
