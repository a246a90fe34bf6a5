"""Mesh -> dense inverse-depth rasterisation, PINNED by the reference's own
fixtures:
  * tests/golden/RasterizationTest_DrawShadedTriangleBarycentric{1,2}.png are the reference's golden
    images (test/data/); test/utils/rasterization_test.cc:514-598 draws one triangle into a 480x640 float
    image, normalises to [0,255], converts to 8 bit and compares BYTE FOR BYTE -- replayed here against
    the CPU checker and against the HIP kernel;
  * ImageUtilsTest.interpolateMeshTest (test/utils/image_utils_test.cc:754-785): known answers at the
    three vertices and the centre of one triangle, tolerance 1e-3.
GPU-vs-checker comparisons on full meshes are bit-exact (NaN pattern included)."""
import os

import numpy as np
import pytest

from flame_amd import synth
from oracle import capi as oracle
from tests.helpers import GOLDEN

CASES = {  # rasterization_test.cc:517-521 and :560-564
    "1": (((150, 100), (100, 200), (200, 300)), (0.0, 1.0, 0.0)),
    "2": (((150, 100), (100, 300), (200, 200)), (0.0, 0.0, 1.0)),
}


def to_u8_like_the_reference_test(img):
    """cv::normalize(img, img, 0, 255, NORM_MINMAX); img.convertTo(out, CV_8U)  (cvRound = half to even)"""
    mn, mx = img.min(), img.max()
    scale = np.float32(255.0) / np.float32(mx - mn)
    n = (img * scale + (np.float32(0) - np.float32(mn) * scale)).astype(np.float32)
    return np.rint(n).clip(0, 255).astype(np.uint8)


def golden(name):
    from PIL import Image

    return np.array(Image.open(os.path.join(GOLDEN, f"RasterizationTest_DrawShadedTriangleBarycentric{name}.png")))


@pytest.mark.parametrize("name", ["1", "2"])
def test_checker_reproduces_reference_golden_png_byte_for_byte(name):
    (p1, p2, p3), (v1, v2, v3) = CASES[name]
    img = np.zeros((480, 640), np.float32)
    oracle.raster_triangle(img, p1, p2, p3, v1, v2, v3)
    assert np.array_equal(to_u8_like_the_reference_test(img), golden(name))


def test_interpolate_mesh_known_answers_from_reference_test():
    tris = np.array([[0, 1, 2]], np.int32)
    vtx = np.array([[10, 10], [20, 10], [10, 20]], np.float32)
    val = np.array([1.0, 2.0, 3.0], np.float32)
    img = oracle.raster_interpolate_mesh(tris, vtx, val, 40, 40)
    assert abs(img[10, 10] - 1.0) < 1e-3 and abs(img[10, 20] - 2.0) < 1e-3 and abs(img[20, 10] - 3.0) < 1e-3
    centre = 0.1 * (15 - 10) + 0.2 * (15 - 10) + 1.0
    assert abs(img[15, 15] - centre) < 1e-3
    assert np.isnan(img[30, 30]) and oracle.raster_coverage(img) == 66  # the 11x11 lower-left triangle


def test_later_triangle_wins_on_shared_pixels():
    """interpolateMesh draws in order: where two triangles overlap the later one stays."""
    vtx = np.array([[10, 10], [30, 10], [10, 30], [25, 28]], np.float32)
    val = np.array([1, 1, 1, 9], np.float32)
    t_a, t_b = [0, 1, 2], [0, 1, 3]  # both contain the pixel (15, 14)
    ab = oracle.raster_interpolate_mesh(np.array([t_a, t_b], np.int32), vtx, val, 40, 40)
    ba = oracle.raster_interpolate_mesh(np.array([t_b, t_a], np.int32), vtx, val, 40, 40)
    only_a = oracle.raster_interpolate_mesh(np.array([t_a], np.int32), vtx, val, 40, 40)
    only_b = oracle.raster_interpolate_mesh(np.array([t_b], np.int32), vtx, val, 40, 40)
    assert not np.isnan(only_a[14, 15]) and not np.isnan(only_b[14, 15]) and only_a[14, 15] != only_b[14, 15]
    assert ab[14, 15] == only_b[14, 15] and ba[14, 15] == only_a[14, 15]


@pytest.fixture(scope="module")
def gpu(built):
    import torch  # noqa: F401

    import flame_amd

    return flame_amd


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["1", "2"])
def test_gpu_reproduces_reference_golden_png_byte_for_byte(gpu, name):
    (p1, p2, p3), (v1, v2, v3) = CASES[name]
    # DrawShadedTriangleBarycentric(p1,p2,p3,...) == interpolateMesh of triangle (2,1,0)
    vtx = np.array([p3, p2, p1], np.float32)
    val = np.array([v3, v2, v1], np.float32)
    with gpu.Regularizer(0) as reg:
        img, cov = reg.interpolate_mesh_arrays(np.array([[0, 1, 2]], np.int32), vtx, val, 480, 640)
    ref = np.full((480, 640), np.nan, np.float32)
    oracle.raster_triangle(ref, p1, p2, p3, v1, v2, v3)
    assert cov == oracle.raster_coverage(ref) and cov >= 7550  # 7550 pixels are > 0 in the golden image
    img = np.where(np.isnan(img), np.float32(0), img)  # the reference test starts from a zero image
    assert np.array_equal(to_u8_like_the_reference_test(img), golden(name))


@pytest.mark.gpu
@pytest.mark.parametrize("config,scale", [("640x480", 1.0), ("1920x1080", 1.37)])
def test_gpu_idepthmap_matches_checker(gpu, config, scale):
    """The real call: rasterise the solver's x*graph_scale over the frame's Delaunay triangles."""
    w, h, _ = synth.CONFIGS[config]
    g = synth.make_graph(config, seed=31)
    tris = synth.delaunay_triangles_scipy(g["pos"])
    with gpu.Regularizer(0) as reg:
        reg.upload_graph(g)
        reg.run(gpu.Params(), 100)
        x = reg.download_state(("x",))["x"]
        img, cov = reg.interpolate_mesh(tris, h, w, graph_scale=scale)
    ref = oracle.raster_interpolate_mesh(tris, g["pos"], (x * np.float32(scale)).astype(np.float32), h, w)
    assert np.array_equal(np.isnan(img), np.isnan(ref))
    m = ~np.isnan(ref)
    assert np.array_equal(img[m], ref[m])
    assert cov == oracle.raster_coverage(ref) and cov > 0.9 * w * h


@pytest.mark.gpu
def test_gpu_validity_masks_and_errors(gpu):
    g = synth.make_graph("320x240", seed=2)
    tris = synth.delaunay_triangles_scipy(g["pos"])
    rng = np.random.default_rng(0)
    tv = (rng.random(len(tris)) > 0.2).astype(np.uint8)
    vv = (rng.random(g["V"]) > 0.05).astype(np.uint8)
    val = g["data_term"]
    with gpu.Regularizer(0) as reg:
        img, cov = reg.interpolate_mesh_arrays(tris, g["pos"], val, 240, 320, vtx_valid=vv, tri_valid=tv)
        ref = oracle.raster_interpolate_mesh(tris, g["pos"], val, 240, 320, tri_valid=tv, vtx_valid=vv)
        assert np.array_equal(np.isnan(img), np.isnan(ref)) and np.array_equal(img[~np.isnan(ref)], ref[~np.isnan(ref)])
        assert cov == oracle.raster_coverage(ref)
        # overlapping triangles: the GPU's atomicMax on (index, value) must keep the later one
        vtx = np.array([[10, 10], [30, 10], [10, 30], [25, 28]], np.float32)
        v4 = np.array([1, 1, 1, 9], np.float32)
        for order in ([[0, 1, 2], [0, 1, 3]], [[0, 1, 3], [0, 1, 2]]):
            got, _ = reg.interpolate_mesh_arrays(np.array(order, np.int32), vtx, v4, 40, 40)
            want = oracle.raster_interpolate_mesh(np.array(order, np.int32), vtx, v4, 40, 40)
            assert np.array_equal(np.isnan(got), np.isnan(want)) and np.array_equal(got[~np.isnan(want)], want[~np.isnan(want)])
        bad = tris.copy()
        bad[0, 0] = g["V"]
        with pytest.raises(gpu.NLTGV2Error):
            reg.interpolate_mesh_arrays(bad, g["pos"], val, 240, 320)
        with pytest.raises(gpu.NLTGV2Error):  # no graph uploaded: the context-bound form needs one
            reg.interpolate_mesh(tris, 240, 320)
