"""The rows of SURVEY.md section 8 chained the way Flame::update() chains them, frame after frame:

    Frame::create -> updateFeatureIDepths -> [projectFeatures: test scaffolding] -> Delaunay -> projectGraph ->
    syncGraph -> NLTGV2-L1 steps -> interpolateMesh

once through the HIP library (C-ABI) and once through the CPU checkers, each side feeding its own outputs
forward.  After every stage of every frame the two sides must hold the same bits; at the end the dense inverse
depth map must also be close to the closed-form truth of the synthetic scene (a slanted textured plane).

The oracle-only variant runs on CPU (no GPU needed) and guards the scaffolding and the physical end result.
"""
import numpy as np
import pytest

from flame_amd import synth
from flame_amd import synth_stereo as ss
from oracle import capi as oracle
from oracle import stereo_capi as so
from oracle import sync_oracle
from tests.helpers import OUT_KEYS, assert_state_equal

W, H, PAD = 320, 240, 5
VAR_MAX_GRAPH = 1e-2  # Params::idepth_var_max_graph (params.h:103)
MARGIN = 8.0
N_ITERS = 60
NEW_FRAMES = (20, 21, 22, 23)


def make_scene():
    sc = ss.PlaneScene(W, H, seed=21, normal=(0.2, -0.1, 1.0), distance=2.2)
    sc.add_camera(10, np.eye(3), [0, 0, 0])
    sc.add_camera(11, ss.rot([0, 1, 0], 0.004), [-0.03, 0.002, -0.005])
    for i, k in enumerate(NEW_FRAMES):
        a = 0.008 + 0.004 * i
        sc.add_camera(k, ss.rot([0.1, 1, 0.05], a), [-0.07 - 0.035 * i, 0.004 + 0.002 * i, -0.015 - 0.01 * i])
    imgs = {c: sc.render(c) for c in sc.cams}
    feats = ss.make_features(sc, so.FEATURE_DTYPE, [10, 11], 500, 21, mu_noise=0.12, var=0.03)
    return sc, imgs, feats


def project_features(sc, feats, k):
    """Scaffolding for Flame::projectFeatures (flame.cc:1754-1860, not a row of section 8): the valid, converged
    features projected into frame k with EpipolarGeometry::project(u, idepth, &u_new, &idepth_new)."""
    sel = np.nonzero((feats["valid"] == 1) & (feats["num_updates"] > 0) & (feats["idepth_var"] < VAR_MAX_GRAPH))[0]
    geos = {a: so.load_geometry(sc.K32, sc.Kinv32, *sc.relative(a, k)) for a in (10, 11)}
    ids, pos, idepth = [], [], []
    for i in sel:
        f = feats[i]
        x, y, d = so.project_idepth(geos[int(f["frame_id"])], float(f["x"]), float(f["y"]), float(f["idepth_mu"]))
        if MARGIN <= x < W - MARGIN and MARGIN <= y < H - MARGIN and d > 0:
            ids.append(int(f["id"]))
            pos.append((x, y))
            idepth.append(d)
    return np.array(ids, np.int32), np.array(pos, np.float32).reshape(-1, 2), np.array(idepth, np.float32)


def projection_between(sc, a, b):
    q, t = sc.relative(a, b)
    Ra, _ = sc.cams[a]
    Rb, _ = sc.cams[b]
    R = (Rb @ Ra.T).astype(np.float32)
    KRKinv = (sc.K32 @ R @ sc.Kinv32).astype(np.float32)
    return q, t, KRKinv


REGION = (MARGIN, MARGIN, W - 2 * MARGIN, H - 2 * MARGIN)


class OracleSide:
    """The CPU checkers chained (oracle/*)."""

    def __init__(self, sc, imgs):
        self.sc, self.imgs = sc, imgs
        self.frames = {}
        self.ref = None
        self.fid = None

    def add_frame(self, k):
        self.frames[k] = so.make_frame(self.imgs[k], PAD)

    def update_features(self, feats, k):
        fr = [dict(p, img_pad=self.frames[p["id"]][0]) for p in ss.poses_for(self.sc, [10, 11], k, 11)]
        rc, st = so.update_feature_idepths(so.Params(), self.sc.K32, self.sc.Kinv32, W, H, PAD, fr, self.frames[k], 11, feats)
        assert rc == 0
        return [int(v) for v in st[:6]]

    def first_graph(self, g, fid):
        self.ref = sync_oracle.RefGraph.from_flat(g, fid)
        self.fid = fid

    def project_graph(self, q, t, KRKinv):
        flat = sync_oracle.flatten(self.ref, self.fid)
        keep = oracle.graph_project(flat["pos"], flat["x"], 1.0, self.sc.K32, self.sc.Kinv32, q, t, KRKinv, REGION)
        sync_oracle.absorb(self.ref, flat, self.fid)
        for i, f in enumerate(self.fid):
            self.ref.v[int(f)]["pos"] = flat["pos"][i].copy()
        return keep

    def sync_graph(self, fid, pos, data, edges):
        sync_oracle.sync(self.ref, fid, pos, data, np.ones(len(fid), np.float32), edges)
        self.fid = fid

    def run(self, n):
        flat = sync_oracle.flatten(self.ref, self.fid)
        assert oracle.run(flat, n) == 0
        sync_oracle.absorb(self.ref, flat, self.fid)

    def state(self):
        return sync_oracle.flatten(self.ref, self.fid)

    def interpolate(self, tris):
        flat = sync_oracle.flatten(self.ref, self.fid)
        return oracle.raster_interpolate_mesh(tris, flat["pos"], flat["x"], H, W)

    def close(self):
        pass


class HipSide:
    """The product: flame_amd.stereo.FeatureTracker + flame_amd.Regularizer over the C-ABI."""

    def __init__(self, sc, imgs):
        import flame_amd
        from flame_amd.stereo import FEATURE_DTYPE, FeatureTracker, StereoParams

        self.sc, self.imgs = sc, imgs
        self.tr = FeatureTracker(sc.K32, sc.Kinv32, W, H, border=PAD)
        self.reg = flame_amd.Regularizer(0)
        self.params = flame_amd.Params()
        self.sp = StereoParams()
        self.dtype = FEATURE_DTYPE

    def add_frame(self, k):
        self.tr.add_frame(k, self.imgs[k])

    def update_features(self, feats, k):
        view = feats.view(self.dtype)
        _, st = self.tr.update_feature_idepths(self.sp, k, 11, ss.poses_for(self.sc, [10, 11], k, 11), view)
        return [st[n] for n in ("num_idepth_updates", "num_fail_max_var", "num_fail_max_dropouts", "num_fail_ref_patch_grad",
                                "num_fail_ambiguous_match", "num_fail_max_cost")]

    def first_graph(self, g, fid):
        self.reg.upload_graph(g)
        self.reg.set_feature_ids(fid)

    def project_graph(self, q, t, KRKinv):
        keep, _ = self.reg.project_graph(self.sc.K32, self.sc.Kinv32, KRKinv, q, t, REGION, graph_scale=1.0)
        return keep

    def sync_graph(self, fid, pos, data, edges):
        self.reg.sync_graph(fid, pos, data, np.ones(len(fid), np.float32), edges)

    def run(self, n):
        self.reg.run(self.params, n)

    def state(self):
        return self.reg.download_state()

    def interpolate(self, tris):
        return self.reg.interpolate_mesh(tris, H, W)[0]

    def close(self):
        self.reg.close()
        self.tr.close()


def drive(sides, sc, imgs, feats0, triangulate):
    """Runs every side through the same frames; asserts bit-identity between sides after every stage."""
    feats = [feats0.copy() for _ in sides]
    for s in sides:
        s.add_frame(10)
        s.add_frame(11)
    prev = None
    dense = None
    for k in NEW_FRAMES:
        stats = []
        for s, f in zip(sides, feats):
            s.add_frame(k)
            stats.append(s.update_features(f, k))
        for st, f in zip(stats[1:], feats[1:]):
            assert st == stats[0], (k, st, stats[0])
            assert f.tobytes() == feats[0].tobytes(), f"frame {k}: features differ after updateFeatureIDepths"
        fid, pos, idepth = project_features(sc, feats[0], k)
        assert len(fid) > 150, (k, len(fid))
        tris, edges = triangulate(pos)
        if prev is None:
            g = synth.assemble_graph(pos, idepth, edges)
            for s in sides:
                s.first_graph(g, fid)
        else:
            q, t, KRKinv = projection_between(sc, prev, k)
            keeps = [s.project_graph(q, t, KRKinv) for s in sides]
            for kp in keeps[1:]:
                assert np.array_equal(kp, keeps[0]), f"frame {k}: projectGraph keep mask"
            for s in sides:
                s.sync_graph(fid, pos, idepth, edges)
        for s in sides:
            s.run(N_ITERS)
        states = [s.state() for s in sides]
        for st in states[1:]:
            assert np.array_equal(st["x"].shape, states[0]["x"].shape)
            assert_state_equal(st, states[0], keys=OUT_KEYS, what=f"frame {k} after {N_ITERS} steps")
        maps = [s.interpolate(tris) for s in sides]
        for m in maps[1:]:
            assert np.array_equal(m, maps[0], equal_nan=True), f"frame {k}: dense inverse depth map"
        prev, dense = k, maps[0]
    return dense, feats[0]


def check_against_truth(sc, dense):
    ys, xs = np.mgrid[0:H, 0:W]
    truth = sc.true_idepth(NEW_FRAMES[-1], np.stack([xs.ravel(), ys.ravel()], 1).astype(np.float32)).reshape(H, W)
    ok = ~np.isnan(dense)
    assert ok.mean() > 0.6, ok.mean()
    rel = np.abs(dense[ok] - truth[ok]) / truth[ok]
    assert np.median(rel) < 0.02 and np.percentile(rel, 90) < 0.08, (np.median(rel), np.percentile(rel, 90))


def _scipy_triangulate(pos):
    return synth.delaunay_triangles_scipy(pos), synth.delaunay_edges_scipy(pos)


def test_checker_pipeline_recovers_the_plane():
    """CPU only: the chained checkers (with Qhull as triangulator) reconstruct the scene."""
    sc, imgs, feats = make_scene()
    side = OracleSide(sc, imgs)
    dense, _ = drive([side], sc, imgs, feats, _scipy_triangulate)
    check_against_truth(sc, dense)


@pytest.mark.gpu
def test_hip_pipeline_matches_checker_pipeline(built):
    import torch  # noqa: F401

    import flame_amd

    sc, imgs, feats = make_scene()
    sides = [OracleSide(sc, imgs), HipSide(sc, imgs)]
    try:
        dense, out = drive(sides, sc, imgs, feats, flame_amd.delaunay)  # the library's own triangulator
        check_against_truth(sc, dense)
        assert (out["num_updates"] == len(NEW_FRAMES)).sum() > 0.5 * len(out)
    finally:
        for s in sides:
            s.close()
