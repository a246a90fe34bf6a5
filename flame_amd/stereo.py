"""ctypes mirror of include/flame_stereo.h: the per-feature epipolar inverse-depth update on MI355X.

Mirrors the reference interface of /root/reference/src/flame/flame.cc:1280-1752 (Flame::updateFeatureIDepths,
Flame::trackFeature) and src/flame/utils/frame.cc:33-71 (Frame::create, level 0).  There is no CPU path: every
call fails with NLTGV2Error when the HIP library or a gfx950 device is missing.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from .regularizer import NLTGV2Error, load_library, status_string

# == flame_stereo_feature == FeatureWithIDepth (flame.h:88-99)
FEATURE_DTYPE = np.dtype([("id", "<u4"), ("frame_id", "<u4"), ("x", "<f4"), ("y", "<f4"), ("idepth_mu", "<f4"),
                          ("idepth_var", "<f4"), ("valid", "u1"), ("reserved_", "u1", (3,)), ("num_updates", "<u4"),
                          ("num_dropouts", "<u4"), ("search_status", "<i4")])
assert FEATURE_DTYPE.itemsize == 40

_PARAM_FIELDS = [("min_baseline", C.c_float), ("do_letterbox", C.c_int32), ("rescale_factor_min", C.c_float),
                 ("rescale_factor_max", C.c_float), ("idepth_var_max", C.c_float), ("max_dropouts", C.c_int32),
                 ("outlier_sigma_thresh", C.c_float), ("do_meas_fusion", C.c_int32), ("win_size", C.c_int32),
                 ("search_sigma", C.c_float), ("min_grad_mag", C.c_float), ("idepth_min", C.c_float),
                 ("idepth_max", C.c_float), ("epilength_min", C.c_float), ("epilength_max", C.c_float),
                 ("process_var_factor", C.c_float), ("process_fail_var_factor", C.c_float), ("max_cost", C.c_float),
                 ("do_subpixel", C.c_int32), ("sample_dist", C.c_float), ("second_best_factor", C.c_float),
                 ("z_win_size", C.c_int32), ("pixel_var", C.c_float), ("epipolar_line_var", C.c_float)]


class StereoParams(C.Structure):
    """flame_stereo_params; defaults are the reference's (flame_stereo_default_params)."""
    _fields_ = _PARAM_FIELDS

    def __init__(self, **kw):
        super().__init__()
        _lib().flame_stereo_default_params(C.byref(self))
        for k, v in kw.items():
            if k not in dict(_PARAM_FIELDS):
                raise TypeError("unknown stereo parameter %r" % k)
            setattr(self, k, v)


class _Pose(C.Structure):
    _fields_ = [("frame_id", C.c_uint32), ("q_ref_to_new", C.c_float * 4), ("t_ref_to_new", C.c_float * 3),
                ("q_ref_to_pf", C.c_float * 4), ("t_ref_to_pf", C.c_float * 3)]


class _Stats(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("num_idepth_updates", "num_fail_max_var", "num_fail_max_dropouts",
                                          "num_fail_ref_patch_grad", "num_fail_ambiguous_match", "num_fail_max_cost",
                                          "success", "error_feature")]


STEREO_ABI_SYMBOLS = (
    "flame_stereo_default_params", "flame_stereo_create", "flame_stereo_destroy", "flame_stereo_set_stream",
    "flame_stereo_set_camera", "flame_stereo_add_frame", "flame_stereo_drop_frame", "flame_stereo_frame_count",
    "flame_stereo_download_frame", "flame_stereo_update_feature_idepths", "flame_stereo_update_feature_idepths_device",
    "flame_stereo_last_kernel_ms", "flame_stereo_last_hip_error", "flame_stereo_set_features",
    "flame_stereo_update_resident", "flame_stereo_get_features", "flame_stereo_features_device", "flame_stereo_set_option",
)
OPT_LANES_PER_FEATURE = 1

_READY = False
_FP = C.POINTER(C.c_float)


def _lib():
    global _READY
    L = load_library()
    if not _READY:
        ctx = C.c_void_p
        PP = C.POINTER(StereoParams)
        sig = {
            "flame_stereo_default_params": (None, [PP]),
            "flame_stereo_create": (C.c_int, [C.POINTER(ctx), C.c_int]),
            "flame_stereo_destroy": (None, [ctx]),
            "flame_stereo_set_stream": (C.c_int, [ctx, C.c_void_p]),
            "flame_stereo_set_camera": (C.c_int, [ctx, _FP, _FP, C.c_int, C.c_int, C.c_int]),
            "flame_stereo_add_frame": (C.c_int, [ctx, C.c_uint32, C.c_void_p, C.c_int]),
            "flame_stereo_drop_frame": (C.c_int, [ctx, C.c_uint32]),
            "flame_stereo_frame_count": (C.c_int, [ctx]),
            "flame_stereo_download_frame": (C.c_int, [ctx, C.c_uint32, C.c_void_p, _FP, _FP]),
            "flame_stereo_update_feature_idepths": (C.c_int, [ctx, PP, C.c_uint32, C.c_uint32, C.c_int, C.POINTER(_Pose),
                                                              C.c_int, C.c_void_p, C.POINTER(_Stats)]),
            "flame_stereo_update_feature_idepths_device": (C.c_int, [ctx, PP, C.c_uint32, C.c_uint32, C.c_int,
                                                                     C.POINTER(_Pose), C.c_int, C.c_void_p,
                                                                     C.POINTER(_Stats)]),
            "flame_stereo_set_features": (C.c_int, [ctx, C.c_int, C.c_void_p]),
            "flame_stereo_update_resident": (C.c_int, [ctx, PP, C.c_uint32, C.c_uint32, C.c_int, C.POINTER(_Pose),
                                                       C.POINTER(_Stats)]),
            "flame_stereo_get_features": (C.c_int, [ctx, C.c_int, C.c_void_p, C.POINTER(C.c_int)]),
            "flame_stereo_features_device": (C.c_int, [ctx, C.POINTER(C.c_void_p), C.POINTER(C.c_int)]),
            "flame_stereo_set_option": (C.c_int, [ctx, C.c_int, C.c_int]),
            "flame_stereo_last_kernel_ms": (C.c_float, [ctx]),
            "flame_stereo_last_hip_error": (C.c_int, [ctx]),
        }
        for name, (res, args) in sig.items():
            fn = getattr(L, name)
            fn.restype, fn.argtypes = res, args
        _READY = True
    return L


def _f32(a, n):
    a = np.ascontiguousarray(a, dtype=np.float32).reshape(-1)
    if a.size != n:
        raise ValueError("expected %d floats" % n)
    return a


class FeatureTracker:
    """Device-side counterpart of the state Flame::updateFeatureIDepths reads: the camera, the resident pose-frames
    (`pfs_`, flame.h:526) and the new frame."""

    def __init__(self, K, Kinv, width: int, height: int, border: int = 5, device: int = 0):
        self._L = _lib()
        self._ctx = C.c_void_p()
        self._chk(self._L.flame_stereo_create(C.byref(self._ctx), device), "create")
        self.width, self.height, self.border = width, height, border
        K, Kinv = _f32(K, 9), _f32(Kinv, 9)
        self._chk(self._L.flame_stereo_set_camera(self._ctx, K.ctypes.data_as(_FP), Kinv.ctypes.data_as(_FP), width, height,
                                                  border), "set_camera")

    def close(self):
        if self._ctx:
            self._L.flame_stereo_destroy(self._ctx)
            self._ctx = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def _chk(self, rc, what):
        if rc != 0:
            raise NLTGV2Error(rc, "%s: %s" % (what, status_string(rc)))

    def add_frame(self, frame_id: int, img: np.ndarray):
        """utils::Frame::create level 0 on the device (frame.cc:33-71)."""
        img = np.ascontiguousarray(img, dtype=np.uint8)
        if img.shape != (self.height, self.width):
            raise ValueError("image must be %dx%d" % (self.height, self.width))
        self._chk(self._L.flame_stereo_add_frame(self._ctx, frame_id, img.ctypes.data, img.strides[0]), "add_frame")

    def drop_frame(self, frame_id: int):
        self._chk(self._L.flame_stereo_drop_frame(self._ctx, frame_id), "drop_frame")

    def frame_count(self) -> int:
        return self._L.flame_stereo_frame_count(self._ctx)

    def download_frame(self, frame_id: int):
        shape = (self.height + 2 * self.border, self.width + 2 * self.border)
        pad = np.empty(shape, np.uint8)
        gx = np.empty(shape, np.float32)
        gy = np.empty(shape, np.float32)
        self._chk(self._L.flame_stereo_download_frame(self._ctx, frame_id, pad.ctypes.data, gx.ctypes.data_as(_FP),
                                                      gy.ctypes.data_as(_FP)), "download_frame")
        return pad, gx, gy

    @staticmethod
    def _poses(poses):
        arr = (_Pose * max(len(poses), 1))()
        for i, p in enumerate(poses):
            arr[i].frame_id = int(p["id"])
            for name, src, n in (("q_ref_to_new", "q_to_new", 4), ("t_ref_to_new", "t_to_new", 3),
                                 ("q_ref_to_pf", "q_to_pf", 4), ("t_ref_to_pf", "t_to_pf", 3)):
                v = _f32(p[src], n)
                for k in range(n):
                    getattr(arr[i], name)[k] = float(v[k])
        return arr

    def update_feature_idepths(self, params: StereoParams, new_frame_id: int, curr_pf_id: int, poses, feats: np.ndarray,
                               raise_on_error: bool = True):
        """Flame::updateFeatureIDepths.  `poses`: list of dicts {id, q_to_new, t_to_new, q_to_pf, t_to_pf};
        `feats` (FEATURE_DTYPE) is updated in place.  Returns (status, stats dict)."""
        if feats.dtype != FEATURE_DTYPE or not feats.flags.c_contiguous:
            raise ValueError("feats must be a contiguous FEATURE_DTYPE array")
        st = _Stats()
        rc = self._L.flame_stereo_update_feature_idepths(self._ctx, C.byref(params), new_frame_id, curr_pf_id, len(poses),
                                                         self._poses(poses), feats.shape[0], feats.ctypes.data, C.byref(st))
        stats = {n: int(getattr(st, n)) for n, _ in _Stats._fields_}
        if rc != 0 and raise_on_error:
            raise NLTGV2Error(rc, "update_feature_idepths: %s (feature %d)" % (status_string(rc), stats["error_feature"]))
        return rc, stats

    def update_feature_idepths_device(self, params: StereoParams, new_frame_id: int, curr_pf_id: int, poses, n_feats: int,
                                      feats_device_ptr: int, wait: bool = True):
        st = _Stats()
        rc = self._L.flame_stereo_update_feature_idepths_device(self._ctx, C.byref(params), new_frame_id, curr_pf_id,
                                                                len(poses), self._poses(poses), n_feats,
                                                                C.c_void_p(feats_device_ptr), C.byref(st) if wait else None)
        self._chk(rc, "update_feature_idepths_device")
        return {n: int(getattr(st, n)) for n, _ in _Stats._fields_} if wait else None

    # ---- the resident feature set (the default way to run the path: features stay on the device between frames) ----
    def set_features(self, feats: np.ndarray):
        if feats.dtype != FEATURE_DTYPE or not feats.flags.c_contiguous:
            raise ValueError("feats must be a contiguous FEATURE_DTYPE array")
        self._chk(self._L.flame_stereo_set_features(self._ctx, feats.shape[0], feats.ctypes.data), "set_features")

    def update_resident(self, params: StereoParams, new_frame_id: int, curr_pf_id: int, poses, wait: bool = True,
                        raise_on_error: bool = True):
        """Flame::updateFeatureIDepths on the resident set, in place on the device.  Returns (status, stats dict)
        (wait=False: enqueued only, stats None)."""
        st = _Stats()
        rc = self._L.flame_stereo_update_resident(self._ctx, C.byref(params), new_frame_id, curr_pf_id, len(poses),
                                                  self._poses(poses), C.byref(st) if wait else None)
        if not wait:
            self._chk(rc, "update_resident")
            return rc, None
        stats = {n: int(getattr(st, n)) for n, _ in _Stats._fields_}
        if rc != 0 and raise_on_error:
            raise NLTGV2Error(rc, "update_resident: %s (feature %d)" % (status_string(rc), stats["error_feature"]))
        return rc, stats

    def get_features(self) -> np.ndarray:
        n = C.c_int(0)
        self._chk(self._L.flame_stereo_get_features(self._ctx, 0, None, C.byref(n)), "get_features")
        out = np.empty(n.value, FEATURE_DTYPE)
        self._chk(self._L.flame_stereo_get_features(self._ctx, n.value, out.ctypes.data, C.byref(n)), "get_features")
        return out

    def features_device(self):
        p, n = C.c_void_p(), C.c_int(0)
        self._chk(self._L.flame_stereo_features_device(self._ctx, C.byref(p), C.byref(n)), "features_device")
        return p.value or 0, n.value

    def set_lanes_per_feature(self, lanes: int):
        self._chk(self._L.flame_stereo_set_option(self._ctx, OPT_LANES_PER_FEATURE, int(lanes)), "set_option")

    def set_stream(self, hip_stream_ptr):
        self._chk(self._L.flame_stereo_set_stream(self._ctx, C.c_void_p(hip_stream_ptr or 0)), "set_stream")

    def last_kernel_ms(self) -> float:
        return float(self._L.flame_stereo_last_kernel_ms(self._ctx))
