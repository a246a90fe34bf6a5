"""tools/cpp_frame_loop.py [--size 640x480] [--frames 12] [--iters 200] [--host-work-us 300] [--keep DIR]
The C++ frame loop (tests/cpp/frame_loop_test.cc: flame_hip::SolverLoop in device mode beside FeatureTracker, delaunayTriangulate,
projectGraph, syncPrepare / syncCommit, interpolateMeshBegin / End) at a BASELINE size: writes the program's input -- the frames, the
poses, the features and per frame the feature set that enters the graph, produced by the library's own tracker through the Python
mirror (bit-identical to the checker: tests/test_stereo.py) -- builds the program and runs it with the test's window open and closed
(`lean`).  The replay of its log on the CPU checkers is tests/test_cpp_facade.py::test_frame_loop_end_to_end (320x240); this tool is
the timing at size.  GPU box."""
import argparse
import os
import struct
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch  # noqa: F401

from flame_amd import synth_stereo as ss
from flame_amd.stereo import FEATURE_DTYPE, FeatureTracker, StereoParams

ap = argparse.ArgumentParser()
ap.add_argument("--size", default="640x480")
ap.add_argument("--frames", type=int, default=12)
ap.add_argument("--iters", type=int, default=200)
ap.add_argument("--host-work-us", type=int, default=300)
ap.add_argument("--keep", default="build/cpp_frame_loop")
a = ap.parse_args()
W, H = [int(v) for v in a.size.split("x")]
PAD, M = 5, 8.0
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sc = ss.PlaneScene(W, H, seed=21, normal=(0.2, -0.1, 1.0), distance=2.2)
sc.add_camera(10, np.eye(3), [0, 0, 0])
sc.add_camera(11, ss.rot([0, 1, 0], 0.004), [-0.03, 0.002, -0.005])
news = list(range(20, 20 + a.frames))
for i, k in enumerate(news):
    sc.add_camera(k, ss.rot([0.1, 1, 0.05], 0.008 + 0.002 * i), [-0.07 - 0.015 * i, 0.004 + 0.001 * i, -0.015 - 0.004 * i])
imgs = {c: sc.render(c) for c in sc.cams}
feats = ss.make_features(sc, FEATURE_DTYPE, [10, 11], (W // 6) * (H // 6) // 2, 21, mu_noise=0.12, var=0.03)


def project_features(f, k):  # (scaffolding for Flame::projectFeatures, as tools/frame_loop.py)
    sel = (f["valid"] == 1) & (f["num_updates"] > 0) & (f["idepth_var"] < 1e-2)
    ids, pos, idp = [], [], []
    for anchor in (10, 11):
        m = sel & (f["frame_id"] == anchor)
        Ra, ta = sc.cams[anchor]
        Rk, tk = sc.cams[k]
        R = Rk @ Ra.T
        t = tk - R @ ta
        u = np.stack([f["x"][m], f["y"][m], np.ones(m.sum())]).astype(np.float64)
        Pc = R @ (np.linalg.inv(sc.K32.astype(np.float64)) @ u / f["idepth_mu"][m].astype(np.float64)) + t[:, None]
        px = sc.K32.astype(np.float64) @ Pc
        x, y = px[0] / px[2], px[1] / px[2]
        ok = (x >= M) & (x < W - M) & (y >= M) & (y < H - M) & (Pc[2] > 0)
        ids.append(f["id"][m][ok]), pos.append(np.stack([x[ok], y[ok]], 1)), idp.append(1.0 / Pc[2][ok])
    return np.concatenate(ids).astype(np.int32), np.concatenate(pos).astype(np.float32), np.concatenate(idp).astype(np.float32)


blob = [struct.pack("<7i", W, H, PAD, len(feats), 2, a.frames, a.host_work_us), sc.K32.astype("<f4").tobytes(), sc.Kinv32.astype("<f4").tobytes()]
for c in (10, 11):
    blob += [struct.pack("<I", c), np.ascontiguousarray(imgs[c], np.uint8).tobytes()]
blob.append(feats.tobytes())
tr = FeatureTracker(sc.K32, sc.Kinv32, W, H, border=PAD)
tr.add_frame(10, imgs[10]), tr.add_frame(11, imgs[11])
f, prev, sizes = feats.copy(), None, []
REGION = (M, M, W - 2 * M, H - 2 * M)
for k in news:
    tr.add_frame(k, imgs[k])
    poses = ss.poses_for(sc, [10, 11], k, 11)
    tr.update_feature_idepths(StereoParams(), k, 11, poses, f)
    tr.drop_frame(k)
    fid, pos, idp = project_features(f, k)
    sizes.append(len(fid))
    blob += [struct.pack("<II", k, 11), np.ascontiguousarray(imgs[k], np.uint8).tobytes(), struct.pack("<i", len(poses))]
    for p in poses:
        blob.append(struct.pack("<I", p["id"]) + np.concatenate([p["q_to_new"], p["t_to_new"], p["q_to_pf"], p["t_to_pf"]]).astype("<f4").tobytes())
    blob += [struct.pack("<i", len(fid)), fid.astype("<i4").tobytes(), pos.astype("<f4").tobytes(), idp.astype("<f4").tobytes()]
    if prev is None:
        blob.append(struct.pack("<i", 0))
    else:
        q, t = sc.relative(prev, k)
        R = (sc.cams[k][0] @ sc.cams[prev][0].T).astype(np.float32)
        KRKinv = (sc.K32 @ R @ sc.Kinv32).astype(np.float32)
        blob.append(struct.pack("<i", 1) + np.concatenate([sc.K32.ravel(), sc.Kinv32.ravel(), KRKinv.ravel(), np.asarray(q, np.float32),
                                                            np.asarray(t, np.float32), np.asarray(REGION, np.float32)]).astype("<f4").tobytes())
    prev = k
tr.close()
keep = os.path.join(ROOT, a.keep)
os.makedirs(keep, exist_ok=True)
fin, exe = os.path.join(keep, "frames.bin"), os.path.join(keep, "frame_loop_test")
with open(fin, "wb") as fh:
    fh.write(b"".join(blob))
lib_dir = os.path.join(ROOT, "flame_amd")
subprocess.check_call(["g++", "-std=c++11", "-O2", "-Wall", "-Wextra", "-Werror", "-Wno-invalid-offsetof", "-pthread", "-I", os.path.join(ROOT, "include"),
                       os.path.join(ROOT, "tests", "cpp", "frame_loop_test.cc"), "-o", exe, "-L", lib_dir, "-lflame_nltgv2_hip",
                       f"-Wl,-rpath,{lib_dir}", "-Wl,-rpath,/opt/rocm/lib"])
print(f"{a.size}: {len(feats)} features, {a.frames} frames, graphs of {min(sizes)}..{max(sizes)} vertices, {a.host_work_us} us of other host work per frame", flush=True)
for lean in ("0", "1"):
    for rep in range(2):
        r = subprocess.run([exe, fin, os.path.join(keep, f"log_{lean}.bin"), str(a.iters), lean], capture_output=True, text=True, timeout=600)
        print((r.stdout + r.stderr).strip(), flush=True)
