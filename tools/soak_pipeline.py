#!/usr/bin/env python3
"""tools/soak_pipeline.py FRAMES [SIZE] -- soak of the persistent solver in PIPELINE conditions (GPU box).

What the reference's process looks like while it runs (flame.cc:99-112 and 302-381): a solver thread iterating on the graph
without pause, and Flame::update() on another thread firing updateFeatureIDepths, the graph sync and interpolateMesh at frame
rate.  Here, for FRAMES frames:

  * solver thread (the SolverLoop contract: a fixed budget per frame, the context touched under one lock): context A runs its
    per-frame budget of iterations in persistent launches (the patch-per-wave kernel; record verification on in two frames of four), while at the same time, on other streams of the same GPU,
  * a tracker thread keeps FeatureTracker.update_resident (the 16-lane epipolar kernel) going back to back, and
  * a raster thread keeps interpolate_mesh_arrays going on a third context;
  * every frame the main thread -- under the lock, as Flame::update does -- reads the solver's mesh out (interpolate_mesh on A),
    edits the graph (8 % churn, re-triangulation by the library's parallel Delaunay) and sync_graph()s it.

Exactness: context B receives the same uploads / syncs and does the same iterations on the ONE-LAUNCH-PER-STEP path with nothing
else running; A's whole state is compared with B's bit for bit every frame (the test-suite ties B's path to the CPU checker),
and for the first CHECK frames both are compared with the chained CPU checkers (oracle/sync_oracle.py + oracle.run).
Reports timeouts_recovered / torn_records_detected / mismatching frames as one JSON line (profiles/r03_soak_pipeline.txt)."""
import json
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch  # noqa: F401  (first: one HIP runtime per process)

import flame_amd
from flame_amd import synth
from flame_amd import synth_stereo as ss
from flame_amd.regularizer import OPT_MESH_STATE, OPT_PERSISTENT, OPT_VERIFY_RECORDS, RUN_PATHS
from flame_amd.stereo import FEATURE_DTYPE, FeatureTracker, StereoParams
from oracle import capi as oracle
from oracle import sync_oracle

FRAMES = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
SIZE = sys.argv[2] if len(sys.argv) > 2 else "640x480"
CHECK = int(os.environ.get("SOAK_CHECK_FRAMES", "12"))  # frames also compared with the chained CPU checkers
ITERS = 200                                              # the per-frame budget: 4 launches of 50
W, H, _ = synth.CONFIGS[SIZE]
KEYS = ("x", "w1", "w2", "x_bar", "w1_bar", "w2_bar", "x_prev", "w1_prev", "w2_prev", "q1", "q2", "q3")
P = flame_amd.Params()
stop = threading.Event()


def _projection():  # a small camera motion (flame.cc:1888-1905): almost every vertex stays inside the region
    K = np.array([[0.82 * W, 0, W / 2.0], [0, 0.82 * W, H / 2.0], [0, 0, 1]], np.float64)
    a = 0.004
    R = np.array([[np.cos(a), 0, np.sin(a)], [0, 1, 0], [-np.sin(a), 0, np.cos(a)]])
    K32, Kinv32 = K.astype(np.float32), np.linalg.inv(K).astype(np.float32)
    KRKinv = (K32 @ R.astype(np.float32) @ Kinv32).astype(np.float32)
    q = np.array([np.cos(a / 2), 0.0, np.sin(a / 2), 0.0], np.float32)
    return K32, Kinv32, KRKinv, q, np.array([0.01, -0.004, 0.006], np.float32), (4.0, 4.0, W - 8.0, H - 8.0)


PROJ = _projection()
load = {"tracker_updates": 0, "raster_calls": 0}


def tracker_load():
    sc = ss.standard_scene(W, H)
    imgs = {c: sc.render(c) for c in (10, 11, 12)}
    feats = ss.make_features(sc, FEATURE_DTYPE, [10, 11], (W // 6) * (H // 6) // 2, 3)
    poses = ss.poses_for(sc, [10, 11], 12, 11)
    with FeatureTracker(sc.K32, sc.Kinv32, W, H) as tr:
        for c, img in imgs.items():
            tr.add_frame(c, img)
        sp = StereoParams()
        while not stop.is_set():
            tr.set_features(feats)
            for _ in range(20):
                tr.update_resident(sp, 12, 11, poses)
                load["tracker_updates"] += 1


def raster_load():
    g = synth.make_graph(SIZE, seed=99)
    tris, _ = synth.delaunay_native(g["pos"])
    with flame_amd.Regularizer(0) as r:
        r.upload_graph(g)
        while not stop.is_set():
            r.interpolate_mesh_arrays(tris, g["pos"], g["data_term"], H, W)
            load["raster_calls"] += 1


def next_frame(rng, feat_id, pos, data, next_id):
    keep = rng.random(len(feat_id)) > 0.08
    feat_id, pos, data = feat_id[keep], pos[keep].copy(), data[keep].copy()
    pos += rng.normal(0, 0.4, pos.shape).astype(np.float32)
    pos[:, 0] = np.clip(pos[:, 0], 1, W - 1)
    pos[:, 1] = np.clip(pos[:, 1], 1, H - 1)
    data = (data + rng.normal(0, 0.01, data.shape)).astype(np.float32)
    n_new = int((~keep).sum())  # as many new features as were lost: the graph keeps its size
    new_pos = np.stack([rng.random(n_new) * (W - 8) + 4, rng.random(n_new) * (H - 8) + 4], 1).astype(np.float32)
    new_data = (0.5 + rng.random(n_new)).astype(np.float32)
    order = rng.permutation(len(feat_id) + n_new)
    feat_id = np.concatenate([feat_id, np.arange(next_id, next_id + n_new)])[order].astype(np.int32)
    return feat_id, np.ascontiguousarray(np.concatenate([pos, new_pos])[order]), np.ascontiguousarray(np.concatenate([data, new_data])[order]), next_id + n_new


def main():
    rng = np.random.default_rng(2026)
    g0 = synth.make_graph(SIZE, seed=31)
    feat_id = np.arange(g0["V"], dtype=np.int32)
    pos, data = g0["pos"].copy(), g0["data_term"].copy()
    next_id = int(feat_id.max()) + 1
    A, B = flame_amd.Regularizer(0), flame_amd.Regularizer(0)
    A.set_option(OPT_PERSISTENT, int(os.environ.get("SOAK_FORM", "4")))  # 4 = the patch-per-wave kernel by name; 1 = the planner's choice, 2 = lane per half-edge
    B.set_option(OPT_PERSISTENT, 0)
    for r in (A, B):
        r.upload_graph(g0)
        r.set_feature_ids(feat_id)
    ref = sync_oracle.RefGraph.from_flat(g0, feat_id) if CHECK else None
    lock = threading.Lock()          # graph_mtx_: the solver thread and update() never touch context A at the same time
    budget = {"left": ITERS, "done": 0}
    frame_ready = threading.Condition(lock)

    def solver_thread():              # SolverLoop::run with max_rounds_per_upload = 4, iters_per_round = 50
        while not stop.is_set():
            with frame_ready:
                while budget["left"] == 0 and not stop.is_set():
                    frame_ready.wait(timeout=0.05)
                if stop.is_set():
                    return
                A.run(P, 50)
                budget["left"] -= 50
                budget["done"] += 50
                frame_ready.notify_all()

    threads = [threading.Thread(target=t, daemon=True) for t in (tracker_load, raster_load, solver_thread)]
    for t in threads:
        t.start()
    mismatches, checker_mismatches, paths, syncs = [], [], {}, {}
    opened = 0
    t0 = time.time()
    tris = synth.delaunay_native(pos)[0]
    for frame in range(FRAMES):
        with frame_ready:             # Flame::update(): wait for the frame's budget, then own the graph
            while budget["left"] > 0:
                frame_ready.wait(timeout=0.05)
            A.set_option(OPT_VERIFY_RECORDS, 1 if (frame & 3) >= 2 else 0)  # (two frames on, two off: an open run applies where it is off)
            pa = RUN_PATHS.get(A.info()["last_run_path"], "?")
            paths[pa] = paths.get(pa, 0) + 1
            B.run(P, ITERS)
            a, b = A.download_state(KEYS), B.download_state(KEYS)
            if not all(np.array_equal(a[k], b[k]) for k in KEYS):
                mismatches.append(frame)
            if frame < CHECK:
                flat = sync_oracle.flatten(ref, feat_id)
                oracle.run(flat, ITERS)
                if not all(np.array_equal(a[k], flat[k]) for k in KEYS):
                    checker_mismatches.append(frame)
                sync_oracle.absorb(ref, flat, feat_id)
            A.interpolate_mesh(tris, H, W)      # the read-back of the frame (flame.cc:372-437)
            feat_prev = feat_id
            feat_id, pos, data, next_id = next_frame(rng, feat_id, pos, data, next_id)
            tris, edges = synth.delaunay_native(pos)
            ones = np.ones(len(feat_id), np.float32)
            if frame & 1:  # round 4: the sync in two halves -- the builder on its side stream while the solver does 50 more iterations on the old graph
                # round 6: what a frame loop's holds enqueue beside / behind rounds in flight -- projectGraph's kernel behind 30 iterations,
                # the commit's expansion beside and its unpack + state gather behind 50 more, the mesh of the committed state beside 10
                it0 = A.iterations()[0]
                if A.run_open(P, 1 << 14):  # an open run: it goes on until projectGraph needs the state (0.2 ms here), then says how far it went
                    time.sleep(0.0002)
                    opened += 1
                else:
                    A.run_async(P, 30)
                keep_a, _ = A.project_graph(*PROJ, graph_scale=1.0)
                n_open = A.iterations()[0] - it0
                B.run(P, n_open)
                keep_b, _ = B.project_graph(*PROJ, graph_scale=1.0)
                if not np.array_equal(keep_a, keep_b):
                    mismatches.append(frame)
                A.sync_prepare(feat_id, pos, data, ones, edges, edges_unique=True)
                A.run_async(P, 50)
                A.sync_commit()
                A.set_option(OPT_MESH_STATE, 1)
                A.run_async(P, 10)
                A.interpolate_mesh_begin(tris, H, W)
                map_a, cov_a = A.interpolate_mesh_end()
                A.set_option(OPT_MESH_STATE, 0)
                B.run(P, 50)
                B.sync_graph(feat_id, pos, data, ones, edges, edges_unique=True)
                map_b, cov_b = B.interpolate_mesh(tris, H, W)
                B.run(P, 10)
                if cov_a != cov_b or not np.array_equal(map_a, map_b, equal_nan=True):
                    mismatches.append(frame)
                if frame < CHECK:
                    flat = sync_oracle.flatten(ref, feat_prev)
                    oracle.run(flat, n_open)
                    oracle.graph_project(flat["pos"], flat["x"], 1.0, PROJ[0], PROJ[1], PROJ[3], PROJ[4], PROJ[2], PROJ[5])
                    oracle.run(flat, 50)
                    sync_oracle.absorb(ref, flat, feat_prev)
            else:
                for r in (A, B):
                    r.sync_graph(feat_id, pos, data, ones, edges, edges_unique=True)  # (the triangulator's own list: the device builder takes it)
            syncs[{1: "host", 2: "device"}.get(A.info()["last_sync_path"], "?")] = syncs.get({1: "host", 2: "device"}.get(A.info()["last_sync_path"], "?"), 0) + 1
            if frame < CHECK:
                sync_oracle.sync(ref, feat_id, pos, data, ones, edges)
                if frame & 1:  # (the 10 iterations beside the mesh)
                    flat = sync_oracle.flatten(ref, feat_id)
                    oracle.run(flat, 10)
                    sync_oracle.absorb(ref, flat, feat_id)
            if (frame + 1) % max(1, FRAMES // 10) == 0:  # (under the lock: a context is one thread's at a time)
                print(f"frame {frame + 1}: {len(mismatches)} mismatching, A recovered {A.info()['timeouts_recovered']} timeouts, "
                      f"{time.time() - t0:.0f} s", flush=True)
            budget["left"] = ITERS
            frame_ready.notify_all()
    stop.set()
    with frame_ready:
        frame_ready.notify_all()
    for t in threads:
        t.join(timeout=10)
    ia = A.info()
    out = {"frames": FRAMES, "size": SIZE, "V_last": int(len(feat_id)), "iterations_per_frame": ITERS, "solver_iterations": budget["done"],
           "solver_run_paths": paths, "sync_paths_of_A": syncs, "open_runs": opened, "every_other_sync": "an open run stopped by projectGraph (its kernel behind the run), sync_prepare + 50 iterations beside the builder + sync_commit (expansion beside, unpack + gather behind), 10 iterations + the mesh of the committed state beside them", "frames_mismatching_the_per_step_reference": len(mismatches), "first_mismatches": mismatches[:5],
           "frames_checked_against_the_chained_cpu_checkers": CHECK, "of_those_mismatching": len(checker_mismatches),
           "timeouts_recovered": int(ia["timeouts_recovered"]), "torn_records_detected": int(ia["torn_records_detected"]),
           "concurrent_load": load, "seconds": round(time.time() - t0, 1),
           "all_ok": not mismatches and not checker_mismatches}
    print(json.dumps(out))
    A.close(), B.close()


if __name__ == "__main__":
    main()
