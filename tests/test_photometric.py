"""BASELINE config 5: per-vertex photometric residual.

CPU: the checker's building blocks against the REFERENCE'S OWN known-answer tests --
  bilinearInterp<uint8_t,float>   /root/reference/test/utils/image_utils_test.cc:150-166
  EpipolarGeometry::project       /root/reference/test/stereo/epipolar_geometry_test.cc:773-806
(these two pieces are therefore pinned; the residual that combines them has no live reference code).
GPU: the HIP epilogue against the checker on a 1920x1080 frame pair, bit for bit."""
import numpy as np
import pytest

from flame_amd import synth
from oracle import capi as oracle


def test_bilinear_known_answers_from_reference_test():
    img = np.array([[91, 210], [162, 95]], np.uint8)  # image_utils_test.cc:151
    for x, y, want in ((0.5, 0.2, 146.1), (0.2, 0.5, 131.70001), (0.5, 0.5, 139.5), (0.2, 0.2, 121.5600052)):
        assert abs(oracle.photo_bilinear_u8(img, x, y) - want) <= 1e-5 * max(1.0, want) + 2e-5  # EXPECT_NEAR(...,1e-5)


def _rot_y(a):
    c, s = np.cos(a), np.sin(a)
    return np.array([[c, 0, s], [0, 1, 0], [-s, 0, c]], np.float64)


def geometry(K, R_ref_to_cmp, t_ref_to_cmp):
    """KRKinv, Kt as EpipolarGeometry::loadGeometry computes them (epipolar_geometry.h:88-93), float32."""
    Kf = K.astype(np.float32)
    Kinv = np.linalg.inv(K).astype(np.float32)
    KRKinv = (Kf @ R_ref_to_cmp.astype(np.float32) @ Kinv).astype(np.float32)
    Kt = (Kf @ t_ref_to_cmp.astype(np.float32)).astype(np.float32)
    return KRKinv, Kt


def test_project_known_answer_from_reference_test():
    """projectTest1: point (1,0,10), T1 = -15 deg yaw at the origin, T2 = translation (1,0,0); projecting
    u2 with idepth 1/10 through T21 must land on u1 within 1e-4 px."""
    K = np.array([[525.0, 0, 320.0], [0, 525.0, 240.0], [0, 0, 1]], np.float64)
    R1, t1 = _rot_y(-np.pi / 12), np.zeros(3)
    R2, t2 = np.eye(3), np.array([1.0, 0, 0])
    p = np.array([1.0, 0.0, 10.0])

    def proj(R, t):  # static project(K,q,t,p), epipolar_geometry.h:114-119
        pc = R.T @ (p - t)
        return (K[0, 0] * pc[0] + K[0, 2] * pc[2]) / pc[2], (K[1, 1] * pc[1] + K[1, 2] * pc[2]) / pc[2]

    u1, u2 = proj(R1, t1), proj(R2, t2)
    # T21 = T1^-1 * T2 : rotation R1^T R2, translation R1^T (t2 - t1)   (ref = camera 2, cmp = camera 1)
    KRKinv, Kt = geometry(K, R1.T @ R2, R1.T @ (t2 - t1))
    cx, cy = oracle.photo_project(KRKinv, Kt, np.float32(u2[0]), np.float32(u2[1]), np.float32(1.0 / p[2]))
    assert abs(cx - u1[0]) < 1e-4 and abs(cy - u1[1]) < 1e-4  # the reference's EXPECT_NEAR(u1, u_cmp, 1e-4)


def smooth_texture(rows, cols, seed):
    rng = np.random.default_rng(seed)
    small = rng.random((rows // 16 + 2, cols // 16 + 2))
    ys = np.linspace(0, small.shape[0] - 1.001, rows)
    xs = np.linspace(0, small.shape[1] - 1.001, cols)
    y0, x0 = ys.astype(int), xs.astype(int)
    fy, fx = (ys - y0)[:, None], (xs - x0)[None, :]
    a = small[y0][:, x0] * (1 - fy) * (1 - fx) + small[y0][:, x0 + 1] * (1 - fy) * fx
    b = small[y0 + 1][:, x0] * fy * (1 - fx) + small[y0 + 1][:, x0 + 1] * fy * fx
    return np.clip((a + b) * 255, 0, 255).astype(np.uint8)


def test_residual_definition_on_cpu():
    """identity geometry and identical images -> residual exactly 0 inside, NaN at the border."""
    ref = smooth_texture(120, 160, 1)
    pos = np.array([[10.5, 20.25], [1.0, 50.0], [80.0, 60.0], [158.5, 60.0]], np.float32)
    x = np.array([0.5, 0.5, 0.0, 0.5], np.float32)
    e = oracle.photo_residual(pos, x, 1.0, np.eye(3, dtype=np.float32), np.zeros(3, np.float32), ref, ref, 3)
    assert e[0] == 0.0 and e[2] == 0.0 and np.isnan(e[1]) and np.isnan(e[3])


@pytest.mark.gpu
def test_gpu_residual_matches_checker_on_1080p(built):
    import torch  # noqa: F401

    import flame_amd

    g = synth.make_graph("1920x1080", seed=55)
    rows, cols = 1080, 1920
    ref, cmp = smooth_texture(rows, cols, 2), smooth_texture(rows, cols, 3)
    K = np.array([[1000.0, 0, 960.0], [0, 1000.0, 540.0], [0, 0, 1]], np.float64)
    KRKinv, Kt = geometry(K, _rot_y(0.01), np.array([0.05, 0.01, 0.002]))  # small-baseline T_ref->cmp
    with flame_amd.Regularizer(0) as reg:
        reg.upload_graph(g)
        reg.run(flame_amd.Params(), 200)
        x = reg.download_state(("x",))["x"]
        reg.photo_set_images(ref, cmp)
        got = reg.photo_residual(KRKinv, Kt, graph_scale=1.3, border=4)
        after = reg.download_state(("x",))["x"]
    want = oracle.photo_residual(g["pos"], x, 1.3, KRKinv, Kt, ref, cmp, 4)
    assert np.array_equal(after, x)  # an epilogue: x untouched
    assert np.array_equal(np.isnan(got), np.isnan(want))
    assert np.isfinite(want).sum() > 0.9 * g["V"]
    assert np.array_equal(got[np.isfinite(want)], want[np.isfinite(want)])


@pytest.mark.gpu
@pytest.mark.parametrize("config,form", [("1920x1080", 1), ("1920x1080", 4), ("1920x1080", 3), ("640x480", 6), ("640x480", 3), ("640x480", 0)])
def test_gpu_residual_fused_into_the_run(built, config, form):
    """BASELINE config 5: the residual of the run's final x produced by the solver's own launch (epilogue of the
    persistent kernels; one appended sweep on the per-step path) equals the stand-alone sweep and the checker."""
    import torch  # noqa: F401

    import flame_amd

    g = synth.make_graph(config, seed=56)
    cols, rows, _ = synth.CONFIGS[config]
    ref, cmp = smooth_texture(rows, cols, 4), smooth_texture(rows, cols, 5)
    K = np.array([[0.52 * cols, 0, cols / 2.0], [0, 0.52 * cols, rows / 2.0], [0, 0, 1]], np.float64)
    KRKinv, Kt = geometry(K, _rot_y(0.008), np.array([0.04, -0.01, 0.003]))
    p = flame_amd.Params()
    with flame_amd.Regularizer(0) as reg:
        reg.set_option(5, form)
        reg.upload_graph(g)
        reg.photo_set_images(ref, cmp)
        reg.photo_fuse(KRKinv, Kt, graph_scale=1.1, border=4)
        for n in (40, 7):
            reg.run(p, n)
            fused = reg.photo_residual_last()
            x = reg.download_state(("x",))["x"]
            want = oracle.photo_residual(g["pos"], x, 1.1, KRKinv, Kt, ref, cmp, 4)
            assert np.array_equal(np.isnan(fused), np.isnan(want)) and np.isfinite(want).sum() > 0.9 * g["V"]
            assert np.array_equal(fused[np.isfinite(want)], want[np.isfinite(want)]), (config, form, n)
            sweep = reg.photo_residual(KRKinv, Kt, graph_scale=1.1, border=4)
            assert np.array_equal(np.nan_to_num(sweep, nan=-1.0), np.nan_to_num(fused, nan=-1.0))
        path = reg.info()["last_run_path"]
        assert (path in (5, 6, 7)) == (form != 0)  # (a persistent form: vertex per lane, patch per wave with one / two half-edges per lane)
        if config == "1920x1080" and form == 1:
            assert path == 7  # BASELINE config 5 runs in the two-half-edges-per-lane patch kernel (12.6 waves per CU)
        if form == 4:
            assert path == 6
        reg.photo_fuse(enable=False)
        with pytest.raises(flame_amd.NLTGV2Error):
            reg.photo_residual_last()
