"""ctypes binding of the stereo part of oracle/liboracle_nltgv2.so (oracle/stereo_oracle.c; TEST INFRASTRUCTURE).

Feature arrays are numpy structured arrays of dtype FEATURE_DTYPE (the layout of flame.h:88-99's
FeatureWithIDepth, 40 bytes).  PARITY: the EpipolarGeometry pieces are pinned by the reference's known-answer
tests (tests/test_stereo.py); the rest is unpinned (see the header of stereo_oracle.c).
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import capi

FEATURE_DTYPE = np.dtype([("id", "<u4"), ("frame_id", "<u4"), ("x", "<f4"), ("y", "<f4"), ("idepth_mu", "<f4"),
                          ("idepth_var", "<f4"), ("valid", "u1"), ("pad_", "u1", (3,)), ("num_updates", "<u4"),
                          ("num_dropouts", "<u4"), ("search_status", "<i4")])
assert FEATURE_DTYPE.itemsize == 40

_FP = C.POINTER(C.c_float)

_PARAM_FIELDS = [("min_baseline", C.c_float, 0.01), ("do_letterbox", C.c_int32, 0),
                 ("rescale_factor_min", C.c_float, 0.7), ("rescale_factor_max", C.c_float, 1.4),
                 ("idepth_var_max", C.c_float, 0.25), ("max_dropouts", C.c_int32, 5),
                 ("outlier_sigma_thresh", C.c_float, 3.0), ("do_meas_fusion", C.c_int32, 1),
                 ("win_size", C.c_int32, 5), ("search_sigma", C.c_float, 2.0), ("min_grad_mag", C.c_float, 5.0),
                 ("idepth_min", C.c_float, 1e-3), ("idepth_max", C.c_float, 2.0), ("epilength_min", C.c_float, 3.0),
                 ("epilength_max", C.c_float, 32.0), ("process_var_factor", C.c_float, 1.01),
                 ("process_fail_var_factor", C.c_float, 1.1), ("max_cost", C.c_float, 1300.0),
                 ("do_subpixel", C.c_int32, 1), ("sample_dist", C.c_float, 1.0), ("second_best_factor", C.c_float, 1.5),
                 ("z_win_size", C.c_int32, 5), ("pixel_var", C.c_float, 16.0), ("epipolar_line_var", C.c_float, 1.0)]


class Params(C.Structure):
    _fields_ = [(n, t) for n, t, _ in _PARAM_FIELDS]

    def __init__(self, **kw):
        super().__init__()
        for n, _, d in _PARAM_FIELDS:
            setattr(self, n, kw.pop(n, d))
        if kw:
            raise TypeError("unknown stereo params: %s" % sorted(kw))


class Geometry(C.Structure):
    _fields_ = [("K", C.c_float * 9), ("Kinv", C.c_float * 9), ("q", C.c_float * 4), ("t", C.c_float * 3),
                ("tcr", C.c_float * 3), ("KRKinv", C.c_float * 9), ("Kt", C.c_float * 3), ("epx", C.c_float),
                ("epy", C.c_float)]


class FrameRef(C.Structure):
    _fields_ = [("id", C.c_uint32), ("img_pad", C.c_void_p), ("q_to_new", C.c_float * 4), ("t_to_new", C.c_float * 3),
                ("q_to_pf", C.c_float * 4), ("t_to_pf", C.c_float * 3)]


_READY = False


def lib():
    global _READY
    L = capi.lib()
    if not _READY:
        GP = C.POINTER(Geometry)
        f = C.c_float
        L.stereo_load_geometry.argtypes = [GP, _FP, _FP, _FP, _FP]
        L.stereo_load_geometry.restype = None
        L.stereo_max_depth_projection.argtypes = [GP, f, f, _FP, _FP]
        L.stereo_max_depth_projection.restype = None
        L.stereo_min_depth_projection.argtypes = [GP, f, f, _FP, _FP]
        L.stereo_min_depth_projection.restype = C.c_int
        L.stereo_project.argtypes = [GP, f, f, f, _FP, _FP]
        L.stereo_project.restype = C.c_int
        L.stereo_project_idepth.argtypes = [GP, f, f, f, _FP, _FP, _FP]
        L.stereo_project_idepth.restype = C.c_int
        L.stereo_epiline.argtypes = [GP, f, f, _FP, _FP, _FP, _FP]
        L.stereo_epiline.restype = C.c_int
        L.stereo_reference_epiline.argtypes = [GP, f, f, _FP, _FP]
        L.stereo_reference_epiline.restype = C.c_int
        L.stereo_disparity.argtypes = [GP, f, f, f, f, _FP, _FP, _FP, _FP, _FP]
        L.stereo_disparity.restype = C.c_int
        L.stereo_disparity_to_depth.argtypes = [GP] + [f] * 7
        L.stereo_disparity_to_depth.restype = f
        L.stereo_disparity_to_idepth.argtypes = [GP] + [f] * 7
        L.stereo_disparity_to_idepth.restype = f
        L.stereo_clip_liang_barsky.argtypes = [f] * 8 + [_FP] * 4
        L.stereo_clip_liang_barsky.restype = C.c_int
        L.stereo_fuse.argtypes = [f, f, f, f, _FP, _FP, f]
        L.stereo_fuse.restype = C.c_int
        L.stereo_update_feature_idepths.argtypes = [C.POINTER(Params), _FP, _FP, C.c_int, C.c_int, C.c_int, C.c_int,
                                                    C.POINTER(FrameRef), C.c_void_p, _FP, _FP, C.c_uint32, C.c_int,
                                                    C.c_void_p, C.POINTER(C.c_int32)]
        L.stereo_update_feature_idepths.restype = C.c_long
        L.stereo_make_frame.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, _FP, _FP]
        L.stereo_make_frame.restype = None
        _READY = True
    return L


def _fp(a):
    return a.ctypes.data_as(_FP)


def _f32(a, n=None):
    a = np.ascontiguousarray(a, dtype=np.float32).reshape(-1)
    assert n is None or a.size == n
    return a


def load_geometry(K, Kinv, q, t) -> Geometry:
    """q = (w, x, y, z) of q_ref_to_cmp."""
    g = Geometry()
    K, Kinv, q, t = _f32(K, 9), _f32(Kinv, 9), _f32(q, 4), _f32(t, 3)
    lib().stereo_load_geometry(C.byref(g), _fp(K), _fp(Kinv), _fp(q), _fp(t))
    return g


def _two(fn, g, *args):
    a, b = C.c_float(), C.c_float()
    rc = fn(C.byref(g), *args, C.byref(a), C.byref(b))
    return rc, np.float32(a.value), np.float32(b.value)


def max_depth_projection(g, ux, uy):
    _, a, b = _two(lib().stereo_max_depth_projection, g, ux, uy)
    return a, b


def min_depth_projection(g, ux, uy):
    rc, a, b = _two(lib().stereo_min_depth_projection, g, ux, uy)
    assert rc == 0
    return a, b


def project(g, ux, uy, idepth):
    rc, a, b = _two(lib().stereo_project, g, ux, uy, idepth)
    assert rc == 0
    return a, b


def project_idepth(g, ux, uy, idepth):
    a, b, c = C.c_float(), C.c_float(), C.c_float()
    rc = lib().stereo_project_idepth(C.byref(g), ux, uy, idepth, C.byref(a), C.byref(b), C.byref(c))
    assert rc == 0
    return np.float32(a.value), np.float32(b.value), np.float32(c.value)


def epiline(g, ux, uy):
    v = [C.c_float() for _ in range(4)]
    rc = lib().stereo_epiline(C.byref(g), ux, uy, *[C.byref(x) for x in v])
    assert rc == 0
    return tuple(np.float32(x.value) for x in v)  # u_inf.x, u_inf.y, epi.x, epi.y


def reference_epiline(g, ux, uy):
    rc, a, b = _two(lib().stereo_reference_epiline, g, ux, uy)
    assert rc == 0
    return a, b


def disparity(g, ux, uy, cx, cy):
    v = [C.c_float() for _ in range(5)]
    rc = lib().stereo_disparity(C.byref(g), ux, uy, cx, cy, *[C.byref(x) for x in v])
    assert rc == 0
    return tuple(np.float32(x.value) for x in v)  # u_inf.x, u_inf.y, epi.x, epi.y, disparity


def disparity_to_depth(g, ux, uy, ix, iy, ex, ey, disp):
    return np.float32(lib().stereo_disparity_to_depth(C.byref(g), ux, uy, ix, iy, ex, ey, disp))


def disparity_to_idepth(g, ux, uy, ix, iy, ex, ey, disp):
    return np.float32(lib().stereo_disparity_to_idepth(C.byref(g), ux, uy, ix, iy, ex, ey, disp))


def clip_liang_barsky(xmin, xmax, ymin, ymax, x0, y0, x1, y1):
    v = [C.c_float() for _ in range(4)]
    rc = lib().stereo_clip_liang_barsky(xmin, xmax, ymin, ymax, x0, y0, x1, y1, *[C.byref(x) for x in v])
    return bool(rc), tuple(np.float32(x.value) for x in v)


def fuse(mu_pred, var_pred, mu_meas, var_meas, thresh=2.0):
    a, b = C.c_float(), C.c_float()
    rc = lib().stereo_fuse(mu_pred, var_pred, mu_meas, var_meas, C.byref(a), C.byref(b), thresh)
    return bool(rc), np.float32(a.value), np.float32(b.value)


def make_frame(img: np.ndarray, border: int):
    """Frame::create level 0: (img_pad u8, gradx_pad f32, grady_pad f32), each (H+2b) x (W+2b)."""
    img = np.ascontiguousarray(img, dtype=np.uint8)
    h, w = img.shape
    shape = (h + 2 * border, w + 2 * border)
    pad = np.empty(shape, np.uint8)
    gx = np.empty(shape, np.float32)
    gy = np.empty(shape, np.float32)
    lib().stereo_make_frame(img.ctypes.data, w, h, border, pad.ctypes.data, _fp(gx), _fp(gy))
    return pad, gx, gy


def update_feature_idepths(params: Params, K, Kinv, width, height, pad, frames, new_frame, curr_pf_id, feats):
    """frames: list of dicts {id, img_pad, q_to_new, t_to_new, q_to_pf, t_to_pf}; new_frame = (img_pad, gradx_pad,
    grady_pad).  `feats` (FEATURE_DTYPE) is updated in place.  Returns (rc, stats[7])."""
    assert feats.dtype == FEATURE_DTYPE and feats.flags.c_contiguous
    K, Kinv = _f32(K, 9), _f32(Kinv, 9)
    arr = (FrameRef * len(frames))()
    keep = []
    for i, fr in enumerate(frames):
        img = np.ascontiguousarray(fr["img_pad"], dtype=np.uint8)
        assert img.shape == (height + 2 * pad, width + 2 * pad)
        keep.append(img)
        arr[i].id = int(fr["id"])
        arr[i].img_pad = img.ctypes.data
        for name, n in (("q_to_new", 4), ("t_to_new", 3), ("q_to_pf", 4), ("t_to_pf", 3)):
            v = _f32(fr[name], n)
            for k in range(n):
                getattr(arr[i], name)[k] = float(v[k])
    ip, gx, gy = (np.ascontiguousarray(new_frame[0], np.uint8), np.ascontiguousarray(new_frame[1], np.float32),
                  np.ascontiguousarray(new_frame[2], np.float32))
    stats = (C.c_int32 * 7)()
    rc = lib().stereo_update_feature_idepths(C.byref(params), _fp(K), _fp(Kinv), width, height, pad, len(frames), arr,
                                             ip.ctypes.data, _fp(gx), _fp(gy), int(curr_pf_id), feats.shape[0],
                                             feats.ctypes.data, stats)
    return int(rc), np.array(list(stats), dtype=np.int32)
