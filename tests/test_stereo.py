"""Per-feature epipolar inverse-depth update.

CPU part: pins oracle/stereo_oracle.c's EpipolarGeometry restatement on the reference's own known-answer
tests (/root/reference/test/stereo/epipolar_geometry_test.cc -- each test below names the one it replays;
values and tolerances are the reference's), and checks the host-side pieces that have no reference test
against independent float64 formulations.
GPU part (-m gpu): the HIP path through the C-ABI against the oracle, bit for bit.
"""
import math

import numpy as np
import pytest

from oracle import stereo_capi as so

K525 = np.array([[525, 0, 320], [0, 525, 240], [0, 0, 1]], np.float32)
KREAL = np.array([[535.43310546875, 0, 320.106652814575], [0, 539.212524414062, 247.632132204719], [0, 0, 1]],
                 np.float32)


def kinv(K):
    return np.linalg.inv(K.astype(np.float64)).astype(np.float32)


def aa(angle, axis):
    """Eigen::Quaternionf(AngleAxisf(angle, axis)) as (w, x, y, z)."""
    h = np.float32(angle) * np.float32(0.5)
    return np.concatenate([[np.cos(h)], np.sin(h) * np.asarray(axis, np.float32)]).astype(np.float32)


def qmul(a, b):
    aw, ax, ay, az = [np.float64(v) for v in a]
    bw, bx, by, bz = [np.float64(v) for v in b]
    return np.array([aw * bw - ax * bx - ay * by - az * bz, aw * bx + ax * bw + ay * bz - az * by,
                     aw * by - ax * bz + ay * bw + az * bx, aw * bz + ax * by - ay * bx + az * bw])


def qconj(q):
    return np.array([q[0], -q[1], -q[2], -q[3]])


def qrot(q, v):
    p = np.concatenate([[0.0], np.asarray(v, np.float64)])
    return qmul(qmul(q, p), qconj(q))[1:]


class SE3:
    """Just enough of Sophus::SE3f for the reference's test scaffolding (float64 here, narrowed on use)."""

    def __init__(self, q, t):
        self.q = np.asarray(q, np.float64)
        self.t = np.asarray(t, np.float64)

    def inverse(self):
        qi = qconj(self.q)
        return SE3(qi, -qrot(qi, self.t))

    def __mul__(self, o):
        if isinstance(o, SE3):
            return SE3(qmul(self.q, o.q), self.t + qrot(self.q, o.t))
        return qrot(self.q, o) + self.t

    def f32(self):
        return self.q.astype(np.float32), self.t.astype(np.float32)


def static_project(K, T, p):
    """EpipolarGeometry::project(K, q, t, p) (epipolar_geometry.h:113-119)."""
    pc = qrot(qconj(T.q), np.asarray(p, np.float64) - T.t)
    return (np.float32((K[0, 0] * pc[0] + K[0, 2] * pc[2]) / pc[2]), np.float32((K[1, 1] * pc[1] + K[1, 2] * pc[2]) / pc[2]))


IDENT = np.array([1, 0, 0, 0], np.float32)


def geo(K, q, t):
    return so.load_geometry(K, kinv(K), q, t)


# ---- reference KATs ------------------------------------------------------------------------------------


def test_kat_min_depth_projection_x_translate_1():  # epipolar_geometry_test.cc:47-71
    x, y = so.min_depth_projection(geo(K525, IDENT, [2, 0, 0]), 320, 240)
    assert x > K525[0, 2] * 2 and y == 240
    x, y = so.min_depth_projection(geo(K525, IDENT, [-2, 0, 0]), 320, 240)
    assert x < 0 and y == 240


def test_kat_min_depth_projection_x_translate_2():  # :76-100
    x, y = so.min_depth_projection(geo(K525, IDENT, [2, 0, 0]), 320, 0)
    assert x > K525[0, 2] * 2 and abs(y) < 1e-3
    x, y = so.min_depth_projection(geo(K525, IDENT, [-2, 0, 0]), 320, 0)
    assert x < 0 and abs(y) < 1e-3


def test_kat_min_depth_projection_y_translate_1():  # :105-129
    x, y = so.min_depth_projection(geo(K525, IDENT, [0, 2, 0]), 320, 240)
    assert y > K525[1, 2] * 2 and abs(x - 320) < 1e-3
    x, y = so.min_depth_projection(geo(K525, IDENT, [0, -2, 0]), 320, 240)
    assert y < 0 and abs(x - 320) < 1e-3


def test_kat_min_depth_projection_y_translate_2():  # :134-158
    x, y = so.min_depth_projection(geo(K525, IDENT, [0, 2, 0]), 0, 240)
    assert y > K525[1, 2] * 2 and abs(x) < 1e-3
    x, y = so.min_depth_projection(geo(K525, IDENT, [0, -2, 0]), 0, 240)
    assert y < 0 and abs(x) < 1e-3


def test_kat_min_depth_projection_60_yaw():  # :163-197
    q21 = aa(-math.pi / 3, [0, 1, 0])
    t21 = np.array([2, 0, 0], np.float64)
    q12 = qconj(q21.astype(np.float64))
    t12 = -qrot(q12, t21)
    x, y = so.min_depth_projection(geo(K525, q12.astype(np.float32), t12.astype(np.float32)), 320, 240)
    assert abs(x - 16.8910904) < 1e-4 and abs(y - 240) < 1e-4
    x, y = so.min_depth_projection(geo(K525, q21, t21), 320, 240)
    assert abs(float(x) - 1049999424) < 1e-4 and abs(y - 240) < 1e-4


def test_kat_min_depth_projection_ref_front_cmp():  # :202-220
    g = geo(KREAL, [0.999138, -0.000878, 0.041493, 0.000386], [-0.221092, -0.036134, 0.084099])
    x, y = so.min_depth_projection(g, 320, 240)
    assert abs(x - -1087.525391) < 1e-2 and abs(y - 15.954912) < 1e-2


def test_kat_min_depth_projection_ref_behind_cmp():  # :225-243
    g = geo(KREAL, [-0.999853, 0.014856, -0.005249, -0.006822], [-0.258187, 0.040849, -0.054990])
    x, y = so.min_depth_projection(g, 320, 240)
    assert abs(x - 187.65597534179688) < 1e-1 and abs(y - 278.55392456054688) < 1e-1


def test_kat_max_depth_projection_identity():  # :248-265
    x, y = so.max_depth_projection(geo(K525, IDENT, [0, 0, 0]), 320, 240)
    assert abs(x - 320) < 1e-3 and abs(y - 240) < 1e-3


def test_kat_max_depth_projection_30_yaw():  # :270-288
    x, y = so.max_depth_projection(geo(K525, aa(-math.pi / 6, [0, 1, 0]), [0, 0, 0]), 320, 240)
    assert abs(x - 16.891090393066406) < 1e-4 and abs(y - 240) < 1e-4


def test_kat_max_depth_projection_30_roll():  # :293-311
    x, y = so.max_depth_projection(geo(K525, aa(-math.pi / 6, [1, 0, 0]), [0, 0, 0]), 320, 240)
    assert abs(x - 320) < 1e-4 and abs(y - 543.10888671875) < 1e-4


def test_kat_epiline_60_yaw():  # :313-336
    qrl = aa(-math.pi / 3, [0, 1, 0])
    qlr = aa(math.pi / 3, [0, 1, 0])
    tlr = -qrot(qrl.astype(np.float64), [2, 0, 0])
    _, _, ex, ey = so.epiline(geo(K525, qlr, tlr.astype(np.float32)), 320, 240)
    assert abs(ex - 1) < 1e-4 and abs(ey) < 1e-4


def test_kat_epiline_60_roll():  # :338-361
    qrl = aa(math.pi / 3, [1, 0, 0])
    qlr = aa(-math.pi / 3, [1, 0, 0])
    tlr = -qrot(qrl.astype(np.float64), [0, 2, 0])
    _, _, ex, ey = so.epiline(geo(K525, qlr, tlr.astype(np.float32)), 320, 240)
    assert abs(ex) < 1e-4 and abs(ey - 1) < 1e-4


DISP_CASES = [  # (yaw of T1, p_world): disparityTo{Depth,InverseDepth}Test1..4 (:370-761)
    (-math.pi / 12, (1.0, 0.0, 10.0)), (-math.pi / 12, (-1.0, 0.0, 10.0)),
    (math.pi / 12, (0.0, 1.0, 10.0)), (math.pi / 12, (0.0, -1.0, 10.0))]


def _two_cameras(yaw, p):
    T1 = SE3(aa(yaw, [0, 1, 0]), [0, 0, 0])
    T2 = SE3(IDENT, [1, 0, 0])
    T12, T21 = T2.inverse() * T1, T1.inverse() * T2
    u1, u2 = static_project(K525, T1, p), static_project(K525, T2, p)
    return T1, T2, T12, T21, u1, u2


@pytest.mark.parametrize("case", range(4))
def test_kat_disparity_to_depth(case):  # :370-561
    yaw, p = DISP_CASES[case]
    T1, T2, T12, T21, u1, u2 = _two_cameras(yaw, p)
    for Tab, Ta, ua, ub in ((T12, T1, u1, u2), (T21, T2, u2, u1)):
        g = geo(K525, *Tab.f32())
        ix, iy, ex, ey, disp = so.disparity(g, ua[0], ua[1], ub[0], ub[1])
        depth = so.disparity_to_depth(g, ua[0], ua[1], ix, iy, ex, ey, disp)
        assert abs(depth - (Ta.inverse() * np.array(p))[2]) < 1e-4


@pytest.mark.parametrize("case", range(4))
def test_kat_disparity_to_inverse_depth(case):  # :570-761
    yaw, p = DISP_CASES[case]
    T1, T2, T12, T21, u1, u2 = _two_cameras(yaw, p)
    for k, (Tab, Ta, ua, ub) in enumerate(((T12, T1, u1, u2), (T21, T2, u2, u1))):
        g = geo(K525, *Tab.f32())
        ix, iy, ex, ey, disp = so.disparity(g, ua[0], ua[1], ub[0], ub[1])
        idepth = so.disparity_to_idepth(g, ua[0], ua[1], ix, iy, ex, ey, disp)
        tol = 1e-2 if (case >= 2 and k == 0) else 1e-4  # the reference loosens tests 3 and 4, camera 1 (:703,753)
        assert abs(idepth - 1.0 / (Ta.inverse() * np.array(p))[2]) < tol


def test_kat_project_1():  # :773-806
    T1, T2, T12, T21, u1, u2 = _two_cameras(-math.pi / 12, (1.0, 0.0, 10.0))
    x, y = so.project(geo(K525, *T21.f32()), u2[0], u2[1], np.float32(1.0 / 10.0))
    assert abs(x - u1[0]) < 1e-4 and abs(y - u1[1]) < 1e-4


# ---- host pieces without reference tests: independent float64 formulations ---------------------------------


def test_project_idepth_matches_matrix_form():
    rng = np.random.default_rng(5)
    for _ in range(50):
        q = rng.normal(size=4)
        q /= np.linalg.norm(q)
        q[0] = abs(q[0]) + 2.0
        q /= np.linalg.norm(q)
        t = rng.normal(size=3) * 0.2
        g = geo(KREAL, q.astype(np.float32), t.astype(np.float32))
        ux, uy, idepth = rng.uniform(20, 600), rng.uniform(20, 440), rng.uniform(0.05, 2.0)
        x1, y1 = so.project(g, ux, uy, idepth)
        x2, y2, nid = so.project_idepth(g, ux, uy, idepth)
        assert abs(x1 - x2) < 2e-2 and abs(y1 - y2) < 2e-2
        p = np.linalg.inv(KREAL.astype(np.float64)) @ np.array([ux, uy, 1.0]) / idepth
        pc = qrot(q, p) + t
        assert abs(nid - 1.0 / pc[2]) < 1e-4 * abs(1.0 / pc[2]) + 1e-6


def test_reference_epiline_points_from_near_to_far():
    # epipolar_geometry.h:295-302: the reference-image epiline of u_ref is the direction in which the
    # projection of a point on the cmp ray moves
    q = aa(0.05, [0, 1, 0])
    t = np.array([0.3, 0.05, 0.02], np.float32)
    g = geo(K525, q, t)
    ex, ey = so.reference_epiline(g, 400, 200)
    assert abs(math.hypot(ex, ey) - 1) < 1e-5
    Tinv = SE3(q, t).inverse()
    c = Tinv.t  # cmp camera centre in the ref frame
    e = np.array([K525[0, 0] * c[0] / c[2] + K525[0, 2], K525[1, 1] * c[1] / c[2] + K525[1, 2]])  # epipole in ref
    d = np.array([400, 200]) - e
    d /= np.linalg.norm(d)
    assert abs(abs(d[0] * ex + d[1] * ey) - 1) < 1e-4


@pytest.mark.parametrize("case", [  # test/utils/image_utils_test.cc:644-752 (LiangBarky*): window, expected clip or None
    ((0, 3, 0, 3), (1.0, 1.0, 2.0, 2.0), (1.0, 1.0, 2.0, 2.0)),
    ((1.1, 3, 1.2, 3), (1.0, 1.0, 2.0, 2.0), (1.2, 1.2, 2.0, 2.0)),
    ((1, 1.5, 1, 1.25), (1.0, 1.0, 2.0, 2.0), (1.0, 1.0, 1.25, 1.25)),
    ((1.1, 1.5, 1.15, 1.25), (1.0, 1.0, 2.0, 2.0), (1.15, 1.15, 1.25, 1.25)),
    ((2 + 1e-3 + 1, 640 - 2 - 1e-3 - 2, 2 + 1e-3 + 1, 480 - 2 - 1e-3 - 2), (-3.40414, 1.95745, -3.26889, 2.17682), None),
])
def test_kat_liang_barsky(case):
    win, seg, want = case
    ok, got = so.clip_liang_barsky(*[np.float32(v) for v in win], *[np.float32(v) for v in seg])
    if want is None:
        assert not ok
    else:
        assert ok and all(abs(float(g) - w) < 1e-6 for g, w in zip(got, want)), (got, want)


@pytest.mark.parametrize("vertical", [False, True])
def test_kat_central_gradient_ramps(vertical):
    """test/utils/image_utils_test.cc:171-326 (getCentralGradientHorizontal/Vertical): a 640x480 ramp image, the
    reference's expectations (central differences inside, one-sided at the borders, tolerance 1e-6)."""
    w, h = 640, 480
    if vertical:
        col = np.floor(np.float32(255.0 / h) * np.arange(h, dtype=np.float32) + np.float32(0.5)).astype(np.uint8)
        img = np.repeat(col[:, None], w, axis=1)
    else:
        row = np.floor(np.float32(255.0 / w) * np.arange(w, dtype=np.float32) + np.float32(0.5)).astype(np.uint8)
        img = np.repeat(row[None, :], h, axis=0)
    _, gx, gy = so.make_frame(np.ascontiguousarray(img), 0)
    f = img.astype(np.float32)
    assert np.abs(gx[:, 1:-1] - 0.5 * (f[:, 2:] - f[:, :-2])).max() < 1e-6
    assert np.abs(gx[:, 0] - (f[:, 1] - f[:, 0])).max() < 1e-6 and np.abs(gx[:, -1] - (f[:, -1] - f[:, -2])).max() < 1e-6
    assert np.abs(gy[1:-1] - 0.5 * (f[2:] - f[:-2])).max() < 1e-6
    assert np.abs(gy[0] - (f[1] - f[0])).max() < 1e-6 and np.abs(gy[-1] - (f[-1] - f[-2])).max() < 1e-6


def test_liang_barsky():
    ok, (a, b, c, d) = so.clip_liang_barsky(1, 639, 1, 479, -10, 100, 700, 100)
    assert ok and (a, b, c, d) == (1, 100, 639, 100)
    ok, _ = so.clip_liang_barsky(1, 639, 1, 479, -10, -5, 700, -5)
    assert not ok
    ok, (a, b, c, d) = so.clip_liang_barsky(1, 639, 1, 479, 10, 10, 20, 30)
    assert ok and (a, b, c, d) == (10, 10, 20, 30)
    ok, _ = so.clip_liang_barsky(1, 639, 1, 479, 700, 10, 800, 30)
    assert not ok


def test_fuse_is_the_product_of_gaussians():
    ok, mu, var = so.fuse(0.5, 0.04, 0.6, 0.01, 3.0)
    assert ok and abs(mu - (0.01 * 0.5 + 0.04 * 0.6) / 0.05) < 1e-6 and abs(var - 0.04 * 0.01 / 0.05) < 1e-7
    ok, _, _ = so.fuse(0.5, 0.0001, 0.6, 0.01, 3.0)  # 10 sigma away
    assert not ok
    ok, mu, var = so.fuse(0.0, 0.25, 0.6, 0.01, 3.0)  # first detection: the measurement
    assert ok and mu == np.float32(0.6) and var == np.float32(0.01)


def test_make_frame_padding_and_gradients():
    rng = np.random.default_rng(2)
    img = rng.integers(0, 256, size=(12, 17), dtype=np.uint8)
    pad, gx, gy = so.make_frame(img, 5)
    assert np.array_equal(pad, np.pad(img, 5, mode="reflect"))  # numpy 'reflect' == cv BORDER_REFLECT_101
    f = img.astype(np.float32)
    ex = np.empty_like(f)
    ex[:, 1:-1] = 0.5 * (f[:, 2:] - f[:, :-2])
    ex[:, 0] = f[:, 1] - f[:, 0]
    ex[:, -1] = f[:, -1] - f[:, -2]
    ey = np.empty_like(f)
    ey[1:-1] = 0.5 * (f[2:] - f[:-2])
    ey[0] = f[1] - f[0]
    ey[-1] = f[-1] - f[-2]
    assert np.array_equal(gx, np.pad(ex, 5)) and np.array_equal(gy, np.pad(ey, 5))


# ---- the per-feature driver on a synthetic plane scene (oracle only; physical sanity of the unpinned parts) ----


def _scene_case(width=320, height=240, n_per_anchor=400, seed=3, **feat_kw):
    from flame_amd import synth_stereo as ss

    sc = ss.standard_scene(width, height, seed=seed)
    imgs = {c: sc.render(c) for c in (10, 11, 12)}
    feats = ss.make_features(sc, so.FEATURE_DTYPE, [10, 11], n_per_anchor, seed, **feat_kw)
    poses = ss.poses_for(sc, [10, 11], 12, 11)
    return sc, imgs, feats, poses


def _oracle_update(sc, imgs, feats, poses, params, pad=5, curr_pf=11, new=12):
    frames = []
    for p in poses:
        frames.append(dict(p, img_pad=so.make_frame(imgs[p["id"]], pad)[0]))
    newf = so.make_frame(imgs[new], pad)
    out = feats.copy()
    rc, stats = so.update_feature_idepths(params, sc.K32, sc.Kinv32, sc.width, sc.height, pad, frames, newf, curr_pf, out)
    return rc, stats, out


def test_oracle_recovers_the_plane():
    sc, imgs, feats, poses = _scene_case()
    rc, stats, out = _oracle_update(sc, imgs, feats, poses, so.Params())
    assert rc == 0
    n = feats.shape[0]
    ok = (out["num_updates"] == 1)
    assert stats[0] == ok.sum() and stats[6] == 1
    assert ok.sum() > 0.6 * n, (ok.sum(), n, stats)
    truth = np.concatenate([sc.true_idepth(a, np.stack([feats["x"], feats["y"]], 1)[feats["frame_id"] == a]) for a in (10, 11)])
    err_prior = np.abs(feats["idepth_mu"][ok] - truth[ok])
    err_post = np.abs(out["idepth_mu"][ok] - truth[ok])
    assert np.median(err_post) < 0.35 * np.median(err_prior), (np.median(err_post), np.median(err_prior))
    assert np.all(out["idepth_var"][ok] < feats["idepth_var"][ok])
    # failures: variance inflated by process_fail_var_factor, dropout counted
    bad = ~ok & (out["frame_id"] == feats["frame_id"])
    assert np.all(out["num_dropouts"][bad] == 1)
    assert np.allclose(out["idepth_var"][bad], feats["idepth_var"][bad] * np.float32(1.1), rtol=1e-6)
    # status counters agree with the per-feature status field
    for k, code in ((3, 1), (4, 2), (5, 3)):
        assert stats[k] == int((out["search_status"] == code).sum())


def test_oracle_small_baseline_skips_everything():
    sc, imgs, feats, poses = _scene_case(n_per_anchor=50)
    for p in poses:
        p["t_to_new"] = (np.asarray(p["t_to_new"]) * 1e-3).astype(np.float32)
    rc, stats, out = _oracle_update(sc, imgs, feats, poses, so.Params())
    assert rc == 0 and not stats.any() and out.tobytes() == feats.tobytes()


def test_oracle_unknown_frame_and_assert_paths():
    sc, imgs, feats, poses = _scene_case(n_per_anchor=50)
    f2 = feats.copy()
    f2["frame_id"][7] = 99
    rc, _, _ = _oracle_update(sc, imgs, f2, poses, so.Params())
    assert rc == 1 + 7  # pfs.at() throws
    f3 = feats.copy()
    f3["idepth_mu"][5] = -0.5  # FLAME_ASSERT(idepth >= 0), epipolar_geometry.h:154
    rc, _, _ = _oracle_update(sc, imgs, f3, poses, so.Params())
    assert rc == -(1 + 5)


# ---- golden fixture (freezes the checker; generator: oracle/make_golden_stereo.py) -----------------------------------


def _load_golden():
    import os

    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "stereo_160x120_s31.npz"))
    poses = [dict(id=int(i), q_to_new=z["q_to_new"][k], t_to_new=z["t_to_new"][k], q_to_pf=z["q_to_pf"][k], t_to_pf=z["t_to_pf"][k])
             for k, i in enumerate(z["pose_ids"])]
    imgs = {10: z["img10"], 11: z["img11"], 12: z["img12"]}
    return z, poses, imgs


def test_checker_reproduces_the_golden_fixture():
    z, poses, imgs = _load_golden()
    w, h, pad = int(z["width"]), int(z["height"]), int(z["pad"])
    feats = np.ascontiguousarray(z["feats_in"]).view(so.FEATURE_DTYPE).reshape(-1).copy()
    frames = [dict(p, img_pad=so.make_frame(imgs[p["id"]], pad)[0]) for p in poses]
    rc, stats = so.update_feature_idepths(so.Params(), z["K"], z["Kinv"], w, h, pad, frames, so.make_frame(imgs[12], pad), 11, feats)
    assert rc == 0 and np.array_equal(stats, z["stats"])
    assert feats.view(np.uint8).reshape(-1, 40).tobytes() == z["feats_out"].tobytes()


# ---- GPU parity: the HIP path through the C-ABI against the oracle, bit for bit ---------------------------------

gpu = pytest.mark.gpu


def _product_params(**kw):
    from flame_amd.stereo import StereoParams

    return StereoParams(**kw)


def _gpu_update(sc, imgs, feats, poses, pkw, pad=5, curr_pf=11, new=12, raise_on_error=True):
    from flame_amd.stereo import FEATURE_DTYPE, FeatureTracker

    with FeatureTracker(sc.K32, sc.Kinv32, sc.width, sc.height, border=pad) as tr:
        for fid, img in imgs.items():
            tr.add_frame(fid, img)
        out = feats.copy().view(FEATURE_DTYPE)
        rc, stats = tr.update_feature_idepths(_product_params(**pkw), new, curr_pf, poses, out, raise_on_error=raise_on_error)
        # the same through the other two ways in: one lane per feature instead of a 16-lane row, and the resident set
        tr.set_lanes_per_feature(1)
        out1 = feats.copy().view(FEATURE_DTYPE)
        rc1, stats1 = tr.update_feature_idepths(_product_params(**pkw), new, curr_pf, poses, out1, raise_on_error=False)
        tr.set_lanes_per_feature(16)
        tr.set_features(feats.copy().view(FEATURE_DTYPE))
        rc2, stats2 = tr.update_resident(_product_params(**pkw), new, curr_pf, poses, raise_on_error=False)
        out2 = tr.get_features()
        assert (rc1, stats1) == (rc, stats) and (rc2, stats2) == (rc, stats), ((rc, stats), (rc1, stats1), (rc2, stats2))
        if rc == 0:
            assert out1.tobytes() == out.tobytes(), "1 lane per feature differs from the 16-lane row"
            assert out2.tobytes() == out.tobytes(), "resident feature set differs from the host-array call"
    return rc, stats, out.view(so.FEATURE_DTYPE)


def _assert_same(sc, imgs, feats, poses, **pkw):
    rc_o, st_o, out_o = _oracle_update(sc, imgs, feats, poses, so.Params(**pkw))
    rc_g, st_g, out_g = _gpu_update(sc, imgs, feats, poses, pkw)
    assert rc_o == 0 and rc_g == 0
    names = ("num_idepth_updates", "num_fail_max_var", "num_fail_max_dropouts", "num_fail_ref_patch_grad",
             "num_fail_ambiguous_match", "num_fail_max_cost", "success")
    assert [st_g[n] for n in names] == [int(v) for v in st_o], (st_g, st_o)
    if out_g.tobytes() != out_o.tobytes():
        for name in out_o.dtype.names:
            a, b = out_g[name], out_o[name]
            bad = np.nonzero((a != b) & ~((a != a) & (b != b)) if a.ndim == 1 else np.any(a != b, axis=1))[0]
            if bad.size:
                i = int(bad[0])
                raise AssertionError("%s differs on %d features, first %d: gpu %r oracle %r (input %r)"
                                     % (name, bad.size, i, a[i], b[i], feats[i]))
    return st_o, out_o


@gpu
def test_gpu_reproduces_the_golden_fixture(built):
    from flame_amd.stereo import FEATURE_DTYPE, FeatureTracker

    z, poses, imgs = _load_golden()
    w, h, pad = int(z["width"]), int(z["height"]), int(z["pad"])
    feats = np.ascontiguousarray(z["feats_in"]).view(FEATURE_DTYPE).reshape(-1).copy()
    with FeatureTracker(z["K"], z["Kinv"], w, h, border=pad) as tr:
        for fid, img in imgs.items():
            tr.add_frame(fid, img)
        rc, st = tr.update_feature_idepths(_product_params(), 12, 11, poses, feats)
    assert rc == 0 and st["num_idepth_updates"] == int(z["stats"][0]) and st["num_fail_max_cost"] == int(z["stats"][5])
    assert feats.view(np.uint8).reshape(-1, 40).tobytes() == z["feats_out"].tobytes()


@gpu
@pytest.mark.parametrize("size", [(320, 240, 400), (640, 480, 4240)])
def test_gpu_update_matches_oracle(built, size):
    sc, imgs, feats, poses = _scene_case(*size)
    st, out = _assert_same(sc, imgs, feats, poses)
    assert st[0] > 0.6 * feats.shape[0]


@gpu
def test_gpu_update_matches_oracle_1080p(built):
    """BASELINE's largest image size: 57 k features anchored in two pose-frames."""
    sc, imgs, feats, poses = _scene_case(1920, 1080, 28800, seed=12)
    st, out = _assert_same(sc, imgs, feats, poses)
    assert feats.shape[0] > 50000 and st[0] > 0.5 * feats.shape[0]


@gpu
def test_gpu_frame_create_matches_oracle(built):
    from flame_amd.stereo import FeatureTracker

    rng = np.random.default_rng(11)
    for (w, h, b) in ((320, 240, 5), (37, 23, 3), (64, 48, 0)):
        img = rng.integers(0, 256, size=(h, w), dtype=np.uint8)
        with FeatureTracker(K525, kinv(K525), w, h, border=b) as tr:
            tr.add_frame(3, img)
            tr.add_frame(4, img[::-1].copy())
            assert tr.frame_count() == 2
            pad, gx, gy = tr.download_frame(3)
            tr.drop_frame(4)
            assert tr.frame_count() == 1
        opad, ogx, ogy = so.make_frame(img, b)
        assert np.array_equal(pad, opad) and np.array_equal(gx, ogx) and np.array_equal(gy, ogy)


@gpu
@pytest.mark.parametrize("pkw", [dict(do_letterbox=1), dict(do_meas_fusion=0), dict(do_subpixel=0),
                                 dict(sample_dist=0.5, epilength_max=48.0), dict(search_sigma=3.0, min_grad_mag=12.0),
                                 dict(max_cost=300.0, second_best_factor=3.0), dict(outlier_sigma_thresh=0.5),
                                 dict(epilength_min=9.0, idepth_min=0.2, idepth_max=0.9), dict(min_baseline=0.1)])
def test_gpu_parameter_variants(built, pkw):
    sc, imgs, feats, poses = _scene_case(320, 240, 500, seed=5)
    _assert_same(sc, imgs, feats, poses, **pkw)


def _camera_case(t_new, R_new=None, n=500, seed=9, **feat_kw):
    from flame_amd import synth_stereo as ss

    sc = ss.PlaneScene(320, 240, seed)
    sc.add_camera(10, np.eye(3), [0, 0, 0])
    sc.add_camera(11, np.eye(3), [-0.03, 0.0, 0.0])
    sc.add_camera(12, np.eye(3) if R_new is None else R_new, t_new)
    imgs = {c: sc.render(c) for c in (10, 11, 12)}
    feats = ss.make_features(sc, so.FEATURE_DTYPE, [10, 11], n, seed, **feat_kw)
    return sc, imgs, feats, ss.poses_for(sc, [10, 11], 12, 11)


@gpu
@pytest.mark.parametrize("t_new", [(-0.1, 0.0, 0.0), (0.0, 0.08, 0.0), (-0.05, 0.0, 0.12), (0.04, -0.02, -0.15)])
def test_gpu_epiline_branches(built, t_new):
    """t_ref_to_cmp.z == 0 (parallel epilines), > 0 (epipole) and < 0 (epipolar_geometry.h:239-264)."""
    sc, imgs, feats, poses = _camera_case(t_new)
    if t_new[2] == 0.0:
        assert poses[0]["t_to_new"][2] == 0.0
    st, _ = _assert_same(sc, imgs, feats, poses)
    assert st[0] > 100


@gpu
def test_gpu_feature_move_to_newest_poseframe(built):
    """Large motion along the optical axis: rescale_factor leaves (0.7, 1.4) and the feature is re-anchored in
    curr_pf (flame.cc:1596-1659) or invalidated when it leaves the image."""
    from flame_amd import synth_stereo as ss

    sc = ss.PlaneScene(320, 240, 4, normal=(0.0, 0.0, 1.0), distance=1.0)
    sc.add_camera(10, np.eye(3), [0, 0, 0])
    sc.add_camera(11, np.eye(3), [0.0, 0.0, -0.33])
    sc.add_camera(12, np.eye(3), [0.01, 0.0, -0.35])
    imgs = {c: sc.render(c) for c in (10, 11, 12)}
    feats = ss.make_features(sc, so.FEATURE_DTYPE, [10, 11], 400, 4)
    poses = ss.poses_for(sc, [10, 11], 12, 11)
    st, out = _assert_same(sc, imgs, feats, poses)
    was10 = feats["frame_id"] == 10
    moved = was10 & (out["frame_id"] == 11)
    assert moved.sum() > 50 and ((out["valid"] == 0) & was10).sum() > 10
    assert np.all(out["num_dropouts"][moved] == 1)


@gpu
def test_gpu_edge_case_features(built):
    sc, imgs, feats, poses = _scene_case(320, 240, 300, seed=8)
    f = feats.copy()
    n = f.shape[0]
    f["idepth_mu"][0:20] = 0.0                      # maxDepthProjection path, var_factor4 = 1
    f["idepth_var"][20:40] = 0.24                   # fails push the variance past idepth_var_max
    f["num_dropouts"][40:60] = 5                    # one more dropout invalidates
    f["x"][60:70] = 2.0                             # outside the valid region (flame.cc:1677)
    f["y"][70:80] = 238.5
    f["idepth_mu"][80:100] *= 3.0                   # far off priors: outliers / failed matches
    f["idepth_var"][100:120] = 1e-6                 # tiny variance: search region padded to epilength_min
    f["idepth_var"][120:140] = 0.0                  # zero variance: empty search segment
    f["search_status"][140:160] = 2                 # stale status of a previous frame is counted again
    f["idepth_mu"][160:170] = 1e-7                  # mu < 1e-6
    f["idepth_mu"][170:180] = 5.0                   # beyond idepth_max: empty search interval
    assert n > 180
    st, out = _assert_same(sc, imgs, f, poses)
    assert (out["valid"][20:60] == 0).any()


@gpu
def test_gpu_sequence_of_frames(built):
    """Three consecutive new frames, the output of one update feeding the next (priors tighten)."""
    from flame_amd import synth_stereo as ss
    from flame_amd.stereo import FEATURE_DTYPE, FeatureTracker

    sc = ss.PlaneScene(320, 240, 6)
    sc.add_camera(10, np.eye(3), [0, 0, 0])
    sc.add_camera(11, ss.rot([0, 1, 0], 0.005), [-0.04, 0.0, -0.01])
    news = {20: ([-0.08, 0.004, -0.02], 0.01), 21: ([-0.12, 0.008, -0.03], 0.015), 22: ([-0.16, 0.01, -0.05], 0.02)}
    for k, (t, a) in news.items():
        sc.add_camera(k, ss.rot([0.1, 1, 0], a), t)
    imgs = {c: sc.render(c) for c in sc.cams}
    feats = ss.make_features(sc, so.FEATURE_DTYPE, [10, 11], 400, 6, mu_noise=0.15, var=0.04)
    fo = feats.copy()
    fg = feats.copy().view(FEATURE_DTYPE)
    with FeatureTracker(sc.K32, sc.Kinv32, sc.width, sc.height) as tr:
        for c in (10, 11):
            tr.add_frame(c, imgs[c])
        for k in news:
            poses = ss.poses_for(sc, [10, 11], k, 11)
            tr.add_frame(k, imgs[k])
            rc, stg = tr.update_feature_idepths(_product_params(), k, 11, poses, fg)
            tr.drop_frame(k)
            frames = [dict(p, img_pad=so.make_frame(imgs[p["id"]], 5)[0]) for p in poses]
            rco, sto = so.update_feature_idepths(so.Params(), sc.K32, sc.Kinv32, sc.width, sc.height, 5, frames,
                                                 so.make_frame(imgs[k], 5), 11, fo)
            assert rc == 0 and rco == 0 and stg["num_idepth_updates"] == sto[0]
            assert fg.tobytes() == fo.tobytes(), k
    truth = np.concatenate([sc.true_idepth(a, np.stack([feats["x"], feats["y"]], 1)[feats["frame_id"] == a]) for a in (10, 11)])
    good = fo["num_updates"] == 3
    assert good.sum() > 0.5 * feats.shape[0]
    assert np.median(np.abs(fo["idepth_mu"][good] - truth[good])) < 0.2 * np.median(np.abs(feats["idepth_mu"][good] - truth[good]))


@gpu
def test_gpu_error_paths(built):
    from flame_amd import NLTGV2Error
    from flame_amd.stereo import FEATURE_DTYPE, FeatureTracker

    sc, imgs, feats, poses = _scene_case(320, 240, 100)
    f = feats.copy()
    f["frame_id"][[31, 7]] = 99
    rc, st, _ = _gpu_update(sc, imgs, f, poses, {}, raise_on_error=False)
    assert rc == -1 and st["error_feature"] == 7         # pfs.at() throws for the first unknown frame
    f = feats.copy()
    f["idepth_mu"][[44, 5]] = -0.5
    rc, st, _ = _gpu_update(sc, imgs, f, poses, {}, raise_on_error=False)
    assert rc == -8 and st["error_feature"] == 5         # FLAME_ASSERT(idepth >= 0)
    assert _oracle_update(sc, imgs, f, poses, so.Params())[0] == -(1 + 5)
    f = feats.copy()
    f["idepth_var"][[60, 23]] = np.float32("nan")       # NaN variance reaches FLAME_ASSERT(!isnan(*mu_post)) in update()
    rc, st, _ = _gpu_update(sc, imgs, f, poses, {}, raise_on_error=False)
    assert rc == -8 and st["error_feature"] == 23
    assert _oracle_update(sc, imgs, f, poses, so.Params())[0] == -(1 + 23)
    with FeatureTracker(sc.K32, sc.Kinv32, sc.width, sc.height) as tr:
        tr.add_frame(10, imgs[10])
        with pytest.raises(NLTGV2Error):                 # new frame not resident
            tr.update_feature_idepths(_product_params(), 12, 11, poses[:1], feats.copy().view(FEATURE_DTYPE))
        with pytest.raises(ValueError):
            tr.add_frame(13, imgs[10][:100])
        e = feats[:0].copy().view(FEATURE_DTYPE)
        tr.add_frame(12, imgs[12])
        rc, st = tr.update_feature_idepths(_product_params(), 12, 11, poses[:1], e)   # empty feature set
        assert rc == 0 and st["success"] == 0 and st["num_idepth_updates"] == 0


@gpu
def test_gpu_device_resident_features(built):
    import torch
    from flame_amd.stereo import FeatureTracker

    sc, imgs, feats, poses = _scene_case(320, 240, 300)
    rc_o, st_o, out_o = _oracle_update(sc, imgs, feats, poses, so.Params())
    dev = torch.from_numpy(feats.view(np.uint8).reshape(-1, 40).copy()).cuda()
    with FeatureTracker(sc.K32, sc.Kinv32, sc.width, sc.height) as tr:
        for fid, img in imgs.items():
            tr.add_frame(fid, img)
        torch.cuda.synchronize()
        st = tr.update_feature_idepths_device(_product_params(), 12, 11, poses, feats.shape[0], dev.data_ptr())
        assert tr.last_kernel_ms() > 0
    assert st["num_idepth_updates"] == st_o[0]
    assert dev.cpu().numpy().tobytes() == out_o.tobytes()
    # the same on a caller-owned stream, enqueue only (no stats): ordered with the caller's other work on that stream
    dev2 = torch.from_numpy(feats.view(np.uint8).reshape(-1, 40).copy()).cuda()
    stream = torch.cuda.Stream()
    with FeatureTracker(sc.K32, sc.Kinv32, sc.width, sc.height) as tr:
        tr.set_stream(stream.cuda_stream)
        for fid, img in imgs.items():
            tr.add_frame(fid, img)
        torch.cuda.synchronize()
        assert tr.update_feature_idepths_device(_product_params(), 12, 11, poses, feats.shape[0], dev2.data_ptr(), wait=False) is None
        with torch.cuda.stream(stream):
            host = dev2.to("cpu", non_blocking=False)
        tr.set_stream(None)
    assert host.numpy().tobytes() == out_o.tobytes()


@gpu
@pytest.mark.parametrize("trial", range(24))
def test_gpu_randomized_differential(built, trial):
    """Random camera motions, plane slants, priors and parameter blocks: the HIP path and the checker must agree on
    every record and counter -- or on the index of the first feature the reference would assert on."""
    from flame_amd import synth_stereo as ss

    rng = np.random.default_rng(1000 + trial)
    w, h = (160, 120) if trial % 3 else (208, 144)
    sc = ss.PlaneScene(w, h, seed=50 + trial, normal=(rng.uniform(-0.4, 0.4), rng.uniform(-0.3, 0.3), 1.0),
                       distance=float(rng.uniform(0.8, 4.0)), margin=96)
    sc.add_camera(10, np.eye(3), [0, 0, 0])
    sc.add_camera(11, ss.rot(rng.normal(size=3), rng.uniform(0, 0.02)), rng.normal(size=3) * 0.03)
    tz = [0.0, rng.uniform(0.02, 0.3), -rng.uniform(0.02, 0.3)][trial % 3]
    t_new = np.array([rng.normal() * 0.08, rng.normal() * 0.04, tz])
    R_new = np.eye(3) if trial % 3 == 0 else ss.rot(rng.normal(size=3), rng.uniform(0, 0.04))
    sc.add_camera(12, R_new, t_new)
    imgs = {c: sc.render(c) for c in (10, 11, 12)}
    feats = ss.make_features(sc, so.FEATURE_DTYPE, [10, 11], 300, 70 + trial, mu_noise=float(rng.uniform(0.02, 0.6)),
                             var=float(rng.choice([1e-4, 0.01, 0.05, 0.2])), border=int(rng.integers(4, 14)))
    n = feats.shape[0]
    # sprinkle awkward priors
    idx = rng.permutation(n)
    feats["idepth_mu"][idx[:10]] = 0.0
    feats["idepth_mu"][idx[10:20]] *= rng.uniform(2.0, 6.0, 10).astype(np.float32)
    feats["idepth_var"][idx[20:30]] = rng.uniform(0.2, 0.26, 10).astype(np.float32)
    feats["num_dropouts"][idx[30:40]] = rng.integers(3, 7, 10).astype(np.uint32)
    feats["search_status"][idx[40:50]] = rng.integers(0, 4, 10).astype(np.int32)
    feats["valid"][idx[50:60]] = 0
    pkw = dict(search_sigma=float(rng.uniform(1.0, 4.0)), min_grad_mag=float(rng.uniform(1.0, 10.0)),
               epilength_min=float(rng.uniform(1.0, 6.0)), epilength_max=float(rng.uniform(8.0, 40.0)),
               max_cost=float(rng.uniform(200.0, 2000.0)), second_best_factor=float(rng.uniform(1.0, 2.5)),
               sample_dist=float(rng.choice([0.5, 1.0, 1.5])), do_subpixel=int(rng.integers(0, 2)),
               do_meas_fusion=int(rng.integers(0, 2)), do_letterbox=int(trial % 5 == 0),
               outlier_sigma_thresh=float(rng.uniform(1.0, 4.0)), idepth_max=float(rng.uniform(1.0, 3.0)),
               rescale_factor_min=float(rng.uniform(0.5, 0.9)), rescale_factor_max=float(rng.uniform(1.1, 1.6)),
               pixel_var=float(rng.uniform(4.0, 32.0)), epipolar_line_var=float(rng.uniform(0.5, 2.0)))
    poses = ss.poses_for(sc, [10, 11], 12, 11)
    rc_o, st_o, out_o = _oracle_update(sc, imgs, feats, poses, so.Params(**pkw))
    rc_g, st_g, out_g = _gpu_update(sc, imgs, feats, poses, pkw, raise_on_error=False)
    if rc_o < 0:
        assert rc_g == -8 and st_g["error_feature"] == -rc_o - 1, (rc_o, rc_g, st_g)
    else:
        assert rc_o == 0 and rc_g == 0, (rc_o, rc_g, st_g)
        assert out_g.tobytes() == out_o.tobytes()
        assert st_g["num_idepth_updates"] == st_o[0] and st_g["num_fail_max_cost"] == st_o[5]


@gpu
def test_gpu_resident_feature_set_api(built):
    """flame_stereo_set_features / update_resident / get_features / features_device: sizes, replacement, argument errors."""
    import ctypes as C

    from flame_amd import NLTGV2Error
    from flame_amd.stereo import FEATURE_DTYPE, FeatureTracker, _lib

    sc, imgs, feats, poses = _scene_case(320, 240, 200)
    f = feats.copy().view(FEATURE_DTYPE)
    with FeatureTracker(sc.K32, sc.Kinv32, sc.width, sc.height, border=5) as tr:
        for fid, img in imgs.items():
            tr.add_frame(fid, img)
        assert tr.get_features().shape == (0,) and tr.features_device() == (0, 0)
        tr.set_features(f)
        ptr, n = tr.features_device()
        assert ptr != 0 and n == f.shape[0]
        assert tr.get_features().tobytes() == f.tobytes()            # round trip before any update
        rc, st = tr.update_resident(_product_params(), 12, 11, poses)
        assert rc == 0 and st["num_idepth_updates"] > 0
        once = tr.get_features()
        rc, _ = tr.update_resident(_product_params(), 12, 11, poses, wait=False)   # enqueued only; ordered on the stream
        twice = tr.get_features()
        assert rc == 0 and (twice["num_updates"] >= once["num_updates"]).all() and twice.tobytes() != once.tobytes()
        tr.set_features(f[:17])                                       # replaced by a smaller set
        assert tr.get_features().shape == (17,)
        L = _lib()
        n_out = C.c_int(0)
        small = np.empty(3, FEATURE_DTYPE)
        assert L.flame_stereo_get_features(tr._ctx, 3, small.ctypes.data, C.byref(n_out)) == -1 and n_out.value == 17
        assert L.flame_stereo_set_features(tr._ctx, -1, None) == -1
        assert L.flame_stereo_set_option(tr._ctx, 1, 7) == -1 and L.flame_stereo_set_option(tr._ctx, 99, 0) == -1
        with pytest.raises(NLTGV2Error):
            tr.update_resident(_product_params(), 12, 77, [dict(p, id=99) for p in poses])  # a pose-frame that is not resident
