"""The chained rows at BASELINE sizes (the frame loop of tests/test_pipeline.py): Frame::create -> updateFeatureIDepths ->
[projectFeatures scaffolding] -> projectGraph -> Delaunay -> syncGraph -> NLTGV2 steps -> interpolateMesh, on the GPU through the C-ABI.

  default        one stage after the other, each timed (what profiles/r0N_frame_loop_*.txt held up to round 3); `--cpu` also times
                 the CPU checkers per stage
  --pipelined    the loop as Flame::update() would drive it with the solver free-running (flame.cc:99-112): the solver iterates in
                 chunks of --iters on its own stream the whole time; projectGraph, the commit of the frame's sync and the start of
                 interpolateMesh are the only points at which it is settled.  Delaunay runs on the host, the sync's builder and the
                 rasteriser on side streams, beside it.  Reported per frame: wall time, solver busy time (HIP events around every
                 chunk), idle = wall - busy, iterations done.
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.getcwd())
import numpy as np
import torch

import flame_amd
from flame_amd import synth
from flame_amd import synth_stereo as ss
from flame_amd.stereo import FEATURE_DTYPE, FeatureTracker, StereoParams

ap = argparse.ArgumentParser()
ap.add_argument("--size", default="640x480")
ap.add_argument("--iters", type=int, default=200)
ap.add_argument("--frames", type=int, default=6)
ap.add_argument("--cpu", action="store_true")
ap.add_argument("--pipelined", action="store_true")
ap.add_argument("--cover", type=float, default=1.25, help="--pipelined: the chunk enqueued before Delaunay + sync_prepare covers this many times their last duration")
ap.add_argument("--host-sync", action="store_true", help="index maps + layout tables on the host (rounds 1-3), for comparison")
ap.add_argument("--open-runs", type=int, default=1, help="--pipelined: 1 = the solver iterates in open runs (flame_nltgv2_run_open) between the calls that need its state; 0 = launches of a guessed length")
ap.add_argument("--mesh-state", type=int, default=1, help="--pipelined: FLAME_NLTGV2_OPT_MESH_STATE; 1 = the mesh of a frame is of the state sync_commit left, begun beside the chunk enqueued behind the commit (0: it settles that chunk first)")
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


def projection(prev, k):
    q, t = sc.relative(prev, k)
    R = (sc.cams[k][0] @ sc.cams[prev][0].T).astype(np.float32)
    return (sc.K32 @ R @ sc.Kinv32).astype(np.float32), q, t


P, SP = flame_amd.Params(), StereoParams()
reg = flame_amd.Regularizer(0)
if a.host_sync:
    reg.set_option(flame_amd.regularizer.OPT_SYNC_PATH, 1)
tr = FeatureTracker(sc.K32, sc.Kinv32, W, H)
tr.add_frame(10, imgs[10]), tr.add_frame(11, imgs[11])
REGION = (M, M, W - 2 * M, H - 2 * M)

if not a.pipelined:
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
            KRKinv, q, t = projection(prev, k)
            t0 = time.perf_counter(); reg.project_graph(sc.K32, sc.Kinv32, KRKinv, q, t, REGION); tick("projectGraph", t0)
            t0 = time.perf_counter(); reg.sync_graph(fid, pos, idp, np.ones(len(fid), np.float32), edges, edges_unique=True); tick("syncGraph", t0)
        t0 = time.perf_counter(); reg.run(P, a.iters); tick("%d NLTGV2 steps" % a.iters, t0)
        t0 = time.perf_counter(); dense, cov = reg.interpolate_mesh(tris, H, W); tick("interpolateMesh (+ D2H of the map)", t0)
        tr.drop_frame(k)
        prev = k
    print("%s: %d features, graph V=%d E=%d, %d updated in the last frame, coverage %.2f, sync path %s" % (
        a.size, len(feats), reg.V, reg.E, st["num_idepth_updates"], cov / (W * H), {1: "host", 2: "device"}.get(reg.info()["last_sync_path"], "-")))
    tot = 0.0
    for name, v in T.items():
        med = float(np.median(v[1:] if len(v) > 2 else v))
        if "first frame" not in name:
            tot += med
        print("  %-42s %8.3f ms (median of %d)" % (name, med, len(v)))
    print("  %-42s %8.3f ms" % ("steady-state frame total", tot))
else:
    # pass 1 (untimed): what the tracker and the scaffolding give every frame, so that the timed loop holds library calls only
    plan, f1 = [], feats.copy()
    for k in news:
        tr.add_frame(k, imgs[k])
        tr.update_feature_idepths(SP, k, 11, ss.poses_for(sc, [10, 11], k, 11), f1)
        plan.append(project_features(f1, k))
        tr.drop_frame(k)
    stream = torch.cuda.Stream(priority=-1)
    reg.set_stream(stream.cuda_stream)
    events, rows = [], []

    labels = []

    iters_done = []

    def solve(ms, label="", closed=False):  # one launch of about `ms` of iterations; label: what stood between the previous launch and this one
        # --open-runs 1: ONE open run instead (flame_nltgv2_run_open: it iterates until the next call that needs the state asks it to stop --
        # no guess at how long the host will take); a call while one is in flight does nothing
        if a.open_runs and not closed and reg.iterations()[1]:
            return
        n = max(a.iters // 4, int(round(ms / chunk_ms * a.iters / 50.0)) * 50)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        before = reg.iterations()[0]
        e0.record(stream)
        if not (a.open_runs and not closed and reg.run_open(P, 1 << 13)):
            reg.run_async(P, n)
        e1.record(stream)
        events.append((e0, e1))
        labels.append(label)
        iters_done.append(before)  # (open runs: the library's count when the launch went out; differences are taken when the loop prints)

    tr.add_frame(news[0], imgs[news[0]])
    tr.update_feature_idepths(SP, news[0], 11, ss.poses_for(sc, [10, 11], news[0], 11), feats)
    tr.drop_frame(news[0])
    fid, pos, idp = plan[0]
    tris, edges = flame_amd.delaunay(pos)
    reg.upload_graph(synth.assemble_graph(pos, idp, edges))
    reg.set_feature_ids(fid)
    reg.run(P, a.iters)
    reg.interpolate_mesh_begin(tris, H, W)
    reg.interpolate_mesh_end(copy=False)
    torch.cuda.synchronize()
    chunk_ms = 1.0
    for _ in range(3):
        solve(0.0, closed=True)
    reg.sync()
    torch.cuda.synchronize()
    chunk_ms = min(e0.elapsed_time(e1) for e0, e1 in events) * 4  # --iters iterations on this graph, in ms (the probe ran iters / 4)
    reg.set_option(flame_amd.regularizer.OPT_MESH_STATE, a.mesh_state)
    events.clear(), labels.clear(), iters_done.clear()
    track_ms, host_ms, build_ms, rast_ms, prev = 0.3, 1.0, 0.3, 0.3, news[0]
    for k, (fid, pos, idp) in zip(news[1:], plan[1:]):
        ones = np.ones(len(fid), np.float32)
        poses = ss.poses_for(sc, [10, 11], k, 11)
        KRKinv, q, t = projection(prev, k)
        n_ev = len(events)
        t_frame = time.perf_counter()
        solve(track_ms, "interpolate_mesh_end -> next frame")
        t0 = time.perf_counter()
        tr.add_frame(k, imgs[k])                                      # Frame::create
        tr.update_feature_idepths(SP, k, 11, poses, feats)            # updateFeatureIDepths
        track_ms = 0.9 * (time.perf_counter() - t0) * 1e3
        reg.project_graph(sc.K32, sc.Kinv32, KRKinv, q, t, REGION)    # settles the solver
        solve(a.cover * host_ms, "Frame::create + updateFeatureIDepths + projectGraph")  # ... which iterates on while the host triangulates
        t0 = time.perf_counter()
        tris, edges = flame_amd.delaunay(pos)
        reg.sync_prepare(fid, pos, idp, ones, edges, edges_unique=True, init_from_map=True, init_graph_scale=1.0)
        host_ms = (time.perf_counter() - t0) * 1e3
        solve(build_ms, "Delaunay + sync_prepare (host)")  # ... and while the builder runs on its side stream
        t0 = time.perf_counter()
        reg.sync_commit()                                             # settles the solver: swap + state gather
        t_commit = (time.perf_counter() - t0) * 1e3
        solve(chunk_ms, "sync_commit")
        # --mesh-state 0: settles the solver; 1: beside the chunk just enqueued, the state the commit left.  Rasteriser + copy-out on a side stream
        reg.interpolate_mesh_begin(tris, H, W)
        solve(max(chunk_ms, 1.05 * rast_ms), "interpolate_mesh_begin")
        t0 = time.perf_counter()
        dense, cov = reg.interpolate_mesh_end(copy=False)
        rast_ms = 0.7 * rast_ms + 0.3 * (rast_ms + (time.perf_counter() - t0) * 1e3 - 0.02)
        tr.drop_frame(k)
        wall = (time.perf_counter() - t_frame) * 1e3
        rows.append((wall, n_ev, len(events), host_ms, t_commit))
        prev = k
    reg.sync()
    torch.cuda.synchronize()
    print("%s pipelined: graph V=%d E=%d, coverage %.2f, sync path %s, %d iterations take %.3f ms" % (
        a.size, reg.V, reg.E, cov / (W * H), {1: "host", 2: "device"}.get(reg.info()["last_sync_path"], "-"), a.iters, chunk_ms))
    print("  frame   wall ms   solver busy ms   idle ms   idle %   iterations   Delaunay + prepare ms   commit call ms")
    tot = []
    for i, (wall, e_a, e_b, host, commit) in enumerate(rows):
        busy = sum(e0.elapsed_time(e1) for e0, e1 in events[e_a:e_b])
        idle = max(0.0, wall - busy)
        n_it = (iters_done[e_b] if e_b < len(iters_done) else reg.iterations()[0]) - iters_done[e_a]
        print("  %5d  %8.3f  %15.3f  %8.3f  %7.1f  %11d  %22.3f  %14.3f" % (i + 1, wall, busy, idle, 100 * idle / wall, n_it, host, commit))
        if i >= 1:
            tot.append((wall, busy, idle, n_it))
    if tot:
        w, b, idl, it = (float(np.median([r[j] for r in tot])) for j in range(4))
        print("  steady state (median, first frame dropped): frame %.3f ms, solver busy %.3f ms, idle %.3f ms = %.1f %%, %d iterations per frame"
              % (w, b, idl, 100 * idl / w, int(it)))
    # where the solver stood still: the gap in front of every chunk (end of the previous chunk -> start of this one, on the device)
    gaps = {}
    first = rows[1][1] if len(rows) > 1 else 0
    for i in range(max(first, 1), len(events)):
        gaps.setdefault(labels[i], []).append(events[i - 1][1].elapsed_time(events[i][0]))
    print("  solver idle by cause (device time between two launches, median over the frames): " +
          ", ".join("%s %.3f ms" % (kk, float(np.median(v))) for kk, v in sorted(gaps.items(), key=lambda kv: -float(np.median(kv[1])))))
    print("  recovered timeouts: %d" % reg.info()["timeouts_recovered"])
if a.cpu and not a.pipelined:
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
