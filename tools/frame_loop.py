"""Per-frame stage timings of the chained rows (the frame loop of tests/test_pipeline.py at BASELINE sizes):
Frame::create -> updateFeatureIDepths -> [projectFeatures scaffolding] -> Delaunay -> projectGraph -> syncGraph ->
N NLTGV2 steps -> interpolateMesh, on the GPU through the C-ABI; `--cpu` also times the CPU checkers per stage."""
import argparse
import os
import sys
import time

sys.path.insert(0, os.getcwd())
import numpy as np
import torch  # noqa: F401

import flame_amd
from flame_amd import synth
from flame_amd import synth_stereo as ss
from flame_amd.stereo import FEATURE_DTYPE, FeatureTracker, StereoParams

ap = argparse.ArgumentParser()
ap.add_argument("--size", default="640x480")
ap.add_argument("--iters", type=int, default=200)
ap.add_argument("--frames", type=int, default=6)
ap.add_argument("--cpu", action="store_true")
a = ap.parse_args()
W, H = [int(v) for v in a.size.split("x")]
sc = ss.PlaneScene(W, H, seed=21, normal=(0.2, -0.1, 1.0), distance=2.2)
sc.add_camera(10, np.eye(3), [0, 0, 0])
sc.add_camera(11, ss.rot([0, 1, 0], 0.004), [-0.03, 0.002, -0.005])
news = list(range(20, 20 + a.frames))
for i, k in enumerate(news):
    sc.add_camera(k, ss.rot([0.1, 1, 0.05], 0.008 + 0.003 * i), [-0.07 - 0.02 * i, 0.004 + 0.001 * i, -0.015 - 0.006 * i])
imgs = {c: sc.render(c) for c in sc.cams}
feats = ss.make_features(sc, FEATURE_DTYPE, [10, 11], (W // 6) * (H // 6) // 2, 21, mu_noise=0.12, var=0.03)
M = 8.0
T = {}


def tick(name, t0):
    T.setdefault(name, []).append((time.perf_counter() - t0) * 1e3)


def project_features(feats, k):  # vectorised scaffolding (float64), not a timed stage
    sel = (feats["valid"] == 1) & (feats["num_updates"] > 0) & (feats["idepth_var"] < 1e-2)
    ids, pos, idp = [], [], []
    for anchor in (10, 11):
        m = sel & (feats["frame_id"] == anchor)
        Ra, ta = sc.cams[anchor]
        Rk, tk = sc.cams[k]
        R = Rk @ Ra.T
        t = tk - R @ ta
        u = np.stack([feats["x"][m], feats["y"][m], np.ones(m.sum())], 0).astype(np.float64)
        P = (np.linalg.inv(sc.K) @ u) / feats["idepth_mu"][m].astype(np.float64)
        Pc = R @ P + t[:, None]
        px = sc.K @ Pc
        x, y = px[0] / px[2], px[1] / px[2]
        ok = (x >= M) & (x < W - M) & (y >= M) & (y < H - M) & (Pc[2] > 0)
        ids.append(feats["id"][m][ok]), pos.append(np.stack([x[ok], y[ok]], 1)), idp.append(1.0 / Pc[2][ok])
    return np.concatenate(ids).astype(np.int32), np.concatenate(pos).astype(np.float32), np.concatenate(idp).astype(np.float32)


P, SP = flame_amd.Params(), StereoParams()
reg = flame_amd.Regularizer(0)
tr = FeatureTracker(sc.K32, sc.Kinv32, W, H)
tr.add_frame(10, imgs[10]), tr.add_frame(11, imgs[11])
prev = None
for k in news:
    t0 = time.perf_counter(); tr.add_frame(k, imgs[k]); tick("Frame::create (upload + pad + gradients)", t0)
    poses = ss.poses_for(sc, [10, 11], k, 11)
    t0 = time.perf_counter(); _, st = tr.update_feature_idepths(SP, k, 11, poses, feats); tick("updateFeatureIDepths (host records)", t0)
    fid, pos, idp = project_features(feats, k)
    t0 = time.perf_counter(); tris, edges = flame_amd.delaunay(pos); tick("Delaunay (host)", t0)
    if prev is None:
        g = synth.assemble_graph(pos, idp, edges)
        t0 = time.perf_counter(); reg.upload_graph(g); reg.set_feature_ids(fid); tick("upload_graph (first frame)", t0)
    else:
        q, t = sc.relative(prev, k)
        R = (sc.cams[k][0] @ sc.cams[prev][0].T).astype(np.float32)
        KRKinv = (sc.K32 @ R @ sc.Kinv32).astype(np.float32)
        t0 = time.perf_counter(); reg.project_graph(sc.K32, sc.Kinv32, KRKinv, q, t, (M, M, W - 2 * M, H - 2 * M)); tick("projectGraph", t0)
        t0 = time.perf_counter(); reg.sync_graph(fid, pos, idp, np.ones(len(fid), np.float32), edges, edges_unique=True); tick("syncGraph", t0)
    t0 = time.perf_counter(); reg.run(P, a.iters); tick("%d NLTGV2 steps" % a.iters, t0)
    t0 = time.perf_counter(); dense, cov = reg.interpolate_mesh(tris, H, W); tick("interpolateMesh (+ D2H of the map)", t0)
    tr.drop_frame(k)
    prev = k
print("%s: %d features, graph V=%d E=%d, %d updated in the last frame, coverage %.2f" % (a.size, len(feats), reg.V, reg.E, st["num_idepth_updates"], cov / (W * H)))
tot = 0.0
for name, v in T.items():
    med = float(np.median(v[1:] if len(v) > 2 else v))
    if "first frame" not in name:
        tot += med
    print("  %-42s %8.3f ms (median of %d)" % (name, med, len(v)))
print("  %-42s %8.3f ms" % ("steady-state frame total", tot))
if a.cpu:
    from oracle import capi as oracle
    from oracle import stereo_capi as so

    k = news[-1]
    t0 = time.perf_counter(); nf = so.make_frame(imgs[k], 5); c_frame = (time.perf_counter() - t0) * 1e3
    frames = [dict(p, img_pad=so.make_frame(imgs[p["id"]], 5)[0]) for p in ss.poses_for(sc, [10, 11], k, 11)]
    f2 = feats.copy().view(so.FEATURE_DTYPE)
    t0 = time.perf_counter(); so.update_feature_idepths(so.Params(), sc.K32, sc.Kinv32, W, H, 5, frames, nf, 11, f2); c_upd = (time.perf_counter() - t0) * 1e3
    g = synth.assemble_graph(pos, idp, edges)
    t0 = time.perf_counter(); oracle.run(g, a.iters); c_run = (time.perf_counter() - t0) * 1e3
    t0 = time.perf_counter(); oracle.raster_interpolate_mesh(tris, pos, g["x"], H, W); c_ras = (time.perf_counter() - t0) * 1e3
    print("  CPU checkers, 1 core: Frame::create %.3f, updateFeatureIDepths %.3f, %d steps %.3f, interpolateMesh %.3f ms"
          % (c_frame, c_upd, a.iters, c_run, c_ras))
reg.close(), tr.close()
