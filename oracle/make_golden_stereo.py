"""Generates tests/golden/stereo_160x120_s31.npz: a small rendered plane scene (three 8-bit frames), the relative poses,
feature records before and after ONE Flame::updateFeatureIDepths as restated by oracle/stereo_oracle.c, and the
counters.  The fixture freezes the checker (it is NOT an output of the reference binary, which cannot be built
here -- see the header of stereo_oracle.c); the EpipolarGeometry part of that checker is pinned separately on the
reference's own known-answer tests (tests/test_stereo.py).

    python -m oracle.make_golden_stereo
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from flame_amd import synth_stereo as ss  # noqa: E402
from oracle import stereo_capi as so  # noqa: E402


def main():
    w, h, pad = 160, 120, 5
    sc = ss.PlaneScene(w, h, seed=31, normal=(0.2, -0.1, 1.0), distance=1.8, margin=64)
    sc.add_camera(10, np.eye(3), [0, 0, 0])
    sc.add_camera(11, ss.rot([0, 1, 0], 0.006), [-0.02, 0.002, -0.004])
    sc.add_camera(12, ss.rot([0.2, 1, 0.1], 0.015), [-0.07, 0.006, -0.03])
    imgs = {c: sc.render(c) for c in (10, 11, 12)}
    feats = ss.make_features(sc, so.FEATURE_DTYPE, [10, 11], 120, 31, mu_noise=0.2, var=0.03)
    feats["idepth_mu"][3] = 0.0
    feats["idepth_var"][5] = 0.24
    feats["num_dropouts"][7] = 5
    feats["idepth_mu"][9] *= np.float32(4.0)
    poses = ss.poses_for(sc, [10, 11], 12, 11)
    frames = [dict(p, img_pad=so.make_frame(imgs[p["id"]], pad)[0]) for p in poses]
    out = feats.copy()
    rc, stats = so.update_feature_idepths(so.Params(), sc.K32, sc.Kinv32, w, h, pad, frames, so.make_frame(imgs[12], pad), 11, out)
    assert rc == 0
    path = os.path.join(ROOT, "tests", "golden", "stereo_160x120_s31.npz")
    np.savez_compressed(
        path, width=w, height=h, pad=pad, K=sc.K32, Kinv=sc.Kinv32, img10=imgs[10], img11=imgs[11], img12=imgs[12],
        pose_ids=np.array([p["id"] for p in poses], np.uint32),
        q_to_new=np.stack([p["q_to_new"] for p in poses]), t_to_new=np.stack([p["t_to_new"] for p in poses]),
        q_to_pf=np.stack([p["q_to_pf"] for p in poses]), t_to_pf=np.stack([p["t_to_pf"] for p in poses]),
        feats_in=feats.view(np.uint8).reshape(-1, 40), feats_out=out.view(np.uint8).reshape(-1, 40), stats=stats)
    print(path, os.path.getsize(path), "bytes;", feats.shape[0], "features, stats", stats.tolist())


if __name__ == "__main__":
    main()
