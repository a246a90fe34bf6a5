"""ctypes mirror of flame::optimizers::nltgv2_l1_graph_regularizer over the C-ABI
(include/flame_nltgv2.h -> libflame_nltgv2_hip.so).

Reference surface (/root/reference/src/flame/optimizers/nltgv2_l1_graph_regularizer.h):
  struct Params h:121-129            -> Params
  step(params, graph) h:134          -> Regularizer.step / .run(n)
  smoothnessCost / dataCost / cost   -> Regularizer.smoothness_cost / data_cost / cost  (h:139-151)
  internal::dualStep / primalStep / extraGradientStep h:158-168
                                     -> Regularizer.dual_step / primal_step / extragradient_step
Graph (h:107-112) is represented by a dict of flat numpy arrays (see flame_amd.synth.assemble_graph):
the order of `src/dst` is boost::edges() order and (src,dst) = (boost::source, boost::target).

This module never computes on the CPU: if the HIP library is missing or no GPU is present the calls
raise NLTGV2Error.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

_FP = C.POINTER(C.c_float)
_IP = C.POINTER(C.c_int32)

# stable options (include/flame_nltgv2.h) ...
OPT_SOLVER, OPT_USE_HIPGRAPH, OPT_PERSISTENT, OPT_PROBE, OPT_VERIFY_RECORDS, OPT_PLACEMENT, OPT_SYNC_PATH, OPT_COST_SUM, OPT_MESH_STATE = 1, 2, 5, 12, 14, 16, 17, 18, 19
# ... and the experimental range (tuning knobs / test hooks of the current kernels; tools/ and the tests use them)
OPT_BLOCK_WAVES, OPT_UNROLL, OPT_DUAL_PUBLISH, OPT_PRESLEEP, OPT_XCDS, OPT_FAULT_INJECT, OPT_POLL_GAP = 103, 104, 106, 108, 109, 110, 113
RUN_PATHS = {0: "none", 1: "persistent (the lane-per-half-edge form, retired in round 3)", 2: "per-step hipGraph", 3: "per-step eager", 4: "canonical 4-sweep",
             5: "persistent-tv", 6: "persistent-pv", 7: "persistent-pv2"}
ERR_NAN = -5

VERTEX_STATE = ("x", "w1", "w2", "x_bar", "w1_bar", "w2_bar", "x_prev", "w1_prev", "w2_prev")
EDGE_STATE = ("q1", "q2", "q3")


class NLTGV2Error(RuntimeError):
    def __init__(self, status: int, what: str):
        super().__init__(f"{what}: status {status} ({status_string(status)})")
        self.status = status


class Params(C.Structure):
    """== struct Params, nltgv2_l1_graph_regularizer.h:121-129 (same defaults)."""

    _fields_ = [(n, C.c_float) for n in ("data_factor", "step_x", "step_q", "theta", "x_min", "x_max")]

    def __init__(self, data_factor=0.1, step_x=0.001, step_q=125.0, theta=0.25, x_min=0.0, x_max=10.0):
        super().__init__(data_factor, step_x, step_q, theta, x_min, x_max)


class _Graph(C.Structure):
    _fields_ = (
        [("V", C.c_int32), ("E", C.c_int32), ("pos", _FP)]
        + [(n, _FP) for n in VERTEX_STATE + ("data_term", "data_weight")]
        + [("src", _IP), ("dst", _IP)]
        + [(n, _FP) for n in ("alpha", "beta") + EDGE_STATE]
    )


class _SyncInput(C.Structure):
    _fields_ = [("V", C.c_int32), ("feat_id", _IP), ("pos", _FP), ("data_term", _FP), ("data_weight", _FP),
                ("init_x", _FP), ("E", C.c_int32), ("edges", _IP), ("check_sticky_obstacles", C.c_int32),
                ("sticky_threshold", C.c_float), ("init_graph_scale", C.c_float), ("edges_unique", C.c_int32),
                ("init_from_map", C.c_int32)]


class _Projection(C.Structure):
    _fields_ = [("K", C.c_float * 9), ("Kinv", C.c_float * 9), ("KRKinv", C.c_float * 9), ("q_ref_to_cmp", C.c_float * 4),
                ("t_ref_to_cmp", C.c_float * 3), ("region_x", C.c_float), ("region_y", C.c_float), ("region_w", C.c_float),
                ("region_h", C.c_float)]


class _Info(C.Structure):
    _fields_ = [
        ("abi_version", C.c_int32), ("device", C.c_int32), ("V", C.c_int32), ("E", C.c_int32),
        ("n_slices", C.c_int32), ("max_degree", C.c_int32), ("padded_half_edges", C.c_int64),
        ("device_bytes", C.c_int64), ("algorithmic_bytes_per_iter", C.c_int64), ("compute_units", C.c_int32),
        ("device_name", C.c_char * 64), ("gcn_arch", C.c_char * 32), ("last_run_path", C.c_int32), ("he_waves", C.c_int32), ("tv_waves", C.c_int32),
        ("tv_wave_capacity", C.c_int32), ("last_run_groups", C.c_int32), ("timeouts_recovered", C.c_int32),
        ("patches", C.c_int32), ("torn_records_detected", C.c_int32), ("last_sync_path", C.c_int32),
        ("last_run_waves_per_cu", C.c_int32), ("reserved0", C.c_int32), ("reserved1", C.c_int32), ("replays_per_step", C.c_int32),
    ]


def library_path() -> str:
    # (FLAME_AMD_LIBRARY: another build of the same library -- same-box A/B of two builds, tools/ab.py)
    return os.environ.get("FLAME_AMD_LIBRARY") or os.path.join(_HERE, "libflame_nltgv2_hip.so")


# every symbol include/flame_nltgv2.h declares (checked by tests/test_abi.py)
ABI_SYMBOLS = (
    "flame_nltgv2_default_params", "flame_nltgv2_create", "flame_nltgv2_destroy", "flame_nltgv2_set_stream", "flame_nltgv2_stream_wait_run", "flame_nltgv2_runs_in_flight", "flame_nltgv2_run_open", "flame_nltgv2_iterations",
    "flame_nltgv2_upload_graph", "flame_nltgv2_update_data", "flame_nltgv2_upload_state", "flame_nltgv2_run",
    "flame_nltgv2_run_async", "flame_nltgv2_sync", "flame_nltgv2_run_timed", "flame_nltgv2_save_prev",
    "flame_nltgv2_dual_step", "flame_nltgv2_primal_step", "flame_nltgv2_extragradient_step", "flame_nltgv2_step",
    "flame_nltgv2_costs", "flame_nltgv2_download_state", "flame_nltgv2_export_idepth_device",
    "flame_nltgv2_export_idepth_device_async", "flame_nltgv2_set_export_target",
    "flame_nltgv2_set_option", "flame_nltgv2_get_info", "flame_nltgv2_last_error", "flame_nltgv2_last_hip_error",
    "flame_nltgv2_status_string", "flame_nltgv2_abi_version", "flame_nltgv2_pack_probe", "flame_nltgv2_read_probe", "flame_nltgv2_layout_selftest", "flame_nltgv2_placement_info",
    "flame_nltgv2_photo_set_images", "flame_nltgv2_photo_residual", "flame_nltgv2_photo_fuse",
    "flame_nltgv2_photo_residual_last", "flame_nltgv2_sync_graph", "flame_nltgv2_sync_prepare", "flame_nltgv2_sync_commit",
    "flame_nltgv2_get_topology", "flame_nltgv2_graph_size", "flame_nltgv2_set_feature_ids", "flame_nltgv2_interpolate_mesh",
    "flame_nltgv2_interpolate_mesh_begin", "flame_nltgv2_interpolate_mesh_end",
    "flame_nltgv2_interpolate_mesh_arrays", "flame_nltgv2_project_graph", "flame_nltgv2_rescale_data",
    "flame_delaunay_triangulate",
)


def load_library():
    """Loads libflame_nltgv2_hip.so (built in-tree by __graft_entry__.build()).  Raises if absent:
    there is no fallback implementation."""
    global _LIB
    if _LIB is not None:
        return _LIB
    path = library_path()
    if not os.path.exists(path):
        raise NLTGV2Error(-2, f"HIP library not built: {path} (run __graft_entry__.build())")
    L = C.CDLL(path)
    ctx = C.c_void_p
    PP, GP = C.POINTER(Params), C.POINTER(_Graph)
    sig = {
        "flame_nltgv2_default_params": (None, [PP]),
        "flame_nltgv2_create": (C.c_int, [C.POINTER(ctx), C.c_int]),
        "flame_nltgv2_destroy": (C.c_int, [ctx]),
        "flame_nltgv2_set_stream": (C.c_int, [ctx, C.c_void_p]),
        "flame_nltgv2_stream_wait_run": (C.c_int, [ctx, C.c_void_p]),
        "flame_nltgv2_runs_in_flight": (C.c_int, [ctx, C.POINTER(C.c_int32)]),
        "flame_nltgv2_run_open": (C.c_int, [ctx, C.POINTER(Params), C.c_int, C.POINTER(C.c_int32)]),
        "flame_nltgv2_iterations": (C.c_int, [ctx, C.POINTER(C.c_int64), C.POINTER(C.c_int32)]),
        "flame_nltgv2_upload_graph": (C.c_int, [ctx, GP]),
        "flame_nltgv2_update_data": (C.c_int, [ctx, _FP, _FP]),
        "flame_nltgv2_upload_state": (C.c_int, [ctx, GP]),
        "flame_nltgv2_run": (C.c_int, [ctx, PP, C.c_int]),
        "flame_nltgv2_run_async": (C.c_int, [ctx, PP, C.c_int]),
        "flame_nltgv2_sync": (C.c_int, [ctx]),
        "flame_nltgv2_run_timed": (C.c_int, [ctx, PP, C.c_int, _FP]),
        "flame_nltgv2_save_prev": (C.c_int, [ctx]),
        "flame_nltgv2_dual_step": (C.c_int, [ctx, PP]),
        "flame_nltgv2_primal_step": (C.c_int, [ctx, PP]),
        "flame_nltgv2_extragradient_step": (C.c_int, [ctx, PP]),
        "flame_nltgv2_step": (C.c_int, [ctx, PP]),
        "flame_nltgv2_costs": (C.c_int, [ctx, PP, _FP, _FP]),
        "flame_nltgv2_download_state": (C.c_int, [ctx, GP]),
        "flame_nltgv2_export_idepth_device": (C.c_int, [ctx, C.c_void_p, C.c_float]),
        "flame_nltgv2_export_idepth_device_async": (C.c_int, [ctx, C.c_void_p, C.c_float]),
        "flame_nltgv2_set_export_target": (C.c_int, [ctx, C.c_void_p, C.c_float]),
        "flame_nltgv2_set_option": (C.c_int, [ctx, C.c_int, C.c_int]),
        "flame_nltgv2_get_info": (C.c_int, [ctx, C.POINTER(_Info)]),
        "flame_nltgv2_last_error": (C.c_int, [ctx]),
        "flame_nltgv2_last_hip_error": (C.c_int, [ctx]),
        "flame_nltgv2_status_string": (C.c_char_p, [C.c_int]),
        "flame_nltgv2_abi_version": (C.c_int, []),
        "flame_nltgv2_layout_selftest": (C.c_int, [ctx, C.POINTER(C.c_int64)]),
        "flame_nltgv2_placement_info": (C.c_int, [ctx, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_float)]),
        "flame_nltgv2_pack_probe": (C.c_int, [GP, _IP, _IP, _IP, _IP, C.c_int64, C.POINTER(C.c_int64)]),
        "flame_nltgv2_read_probe": (C.c_int, [ctx, C.POINTER(C.c_uint32), C.c_int64, C.POINTER(C.c_int64)]),
        "flame_nltgv2_photo_set_images": (C.c_int, [ctx, C.POINTER(C.c_uint8), C.POINTER(C.c_uint8), C.c_int, C.c_int, C.c_int]),
        "flame_nltgv2_photo_residual": (C.c_int, [ctx, _FP, _FP, C.c_float, C.c_int, _FP]),
        "flame_nltgv2_photo_fuse": (C.c_int, [ctx, _FP, _FP, C.c_float, C.c_int, C.c_int]),
        "flame_nltgv2_photo_residual_last": (C.c_int, [ctx, _FP]),
        "flame_nltgv2_sync_graph": (C.c_int, [ctx, C.POINTER(_SyncInput)]),
        "flame_nltgv2_sync_prepare": (C.c_int, [ctx, C.POINTER(_SyncInput)]),
        "flame_nltgv2_sync_commit": (C.c_int, [ctx]),
        "flame_nltgv2_get_topology": (C.c_int, [ctx, _IP, _IP, _IP]),
        "flame_nltgv2_graph_size": (C.c_int, [ctx, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
        "flame_nltgv2_set_feature_ids": (C.c_int, [ctx, _IP]),
        "flame_delaunay_triangulate": (C.c_int, [_FP, C.c_int32, _IP, C.c_int32, _IP, _IP, C.c_int32, _IP]),
        "flame_nltgv2_project_graph": (C.c_int, [ctx, C.POINTER(_Projection), C.c_float, C.POINTER(C.c_uint8), _FP]),
        "flame_nltgv2_rescale_data": (C.c_int, [ctx, C.c_float, _FP, PP]),
        "flame_nltgv2_interpolate_mesh": (C.c_int, [ctx, _IP, C.c_int32, C.POINTER(C.c_uint8), C.c_int, C.c_int, C.c_float,
                                                    _FP, _IP]),
        "flame_nltgv2_interpolate_mesh_begin": (C.c_int, [ctx, _IP, C.c_int32, C.POINTER(C.c_uint8), C.c_int, C.c_int, C.c_float]),
        "flame_nltgv2_interpolate_mesh_end": (C.c_int, [ctx, C.POINTER(_FP), _IP]),
        "flame_nltgv2_interpolate_mesh_arrays": (C.c_int, [ctx, _IP, C.c_int32, _FP, _FP, C.c_int32, C.POINTER(C.c_uint8),
                                                           C.POINTER(C.c_uint8), C.c_int, C.c_int, _FP, _IP]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(L, name)
        fn.restype, fn.argtypes = res, args
    _LIB = L
    return L


def status_string(status: int) -> str:
    try:
        return load_library().flame_nltgv2_status_string(int(status)).decode()
    except Exception:  # library missing
        return "?"


def _as(arr, dtype, n, name):
    a = np.ascontiguousarray(arr, dtype=dtype)
    if a.size != n:
        raise ValueError(f"{name}: expected {n} elements, got {a.size}")
    return a


def _graph_view(g: dict, keep: list, need_all=True) -> _Graph:
    V, E = int(g["V"]), int(g["E"])
    cg = _Graph()
    cg.V, cg.E = V, E
    sizes = {"pos": 2 * V, "data_term": V, "data_weight": V, "src": E, "dst": E, "alpha": E, "beta": E}
    sizes.update({k: V for k in VERTEX_STATE})
    sizes.update({k: E for k in EDGE_STATE})
    for name, ctype in _Graph._fields_[2:]:
        if name not in g or g[name] is None:
            if need_all and not name.endswith("_prev"):
                raise KeyError(name)
            continue
        dt = np.int32 if ctype is _IP else np.float32
        a = _as(g[name], dt, sizes[name], name)
        keep.append(a)
        setattr(cg, name, a.ctypes.data_as(ctype))
    return cg


def delaunay(pos, out=None):
    """Delaunay triangulation of float32 points with the library's own triangulator (host code, exact
    predicates).  Returns (triangles (T,3) int32 counter-clockwise, edges (E,2) int32).  `out` = (triangles, edges) arrays of at least
    (2 n, 3) and (3 n, 2) int32 to write into -- what a frame loop keeps between frames, as the C++ facade's vectors do (fresh arrays
    of this size are fresh pages: the library's writers fault them in)."""
    L = load_library()
    p = np.ascontiguousarray(pos, np.float32).reshape(-1, 2)
    n = p.shape[0]
    nt, ne = C.c_int32(0), C.c_int32(0)
    if out is not None and out[0].shape[0] >= 2 * n and out[1].shape[0] >= 3 * n and out[0].dtype == np.int32 and out[1].dtype == np.int32:
        tri, edg = out
    else:
        tri = np.empty((max(2 * n, 1), 3), np.int32)
        edg = np.empty((max(3 * n, 1), 2), np.int32)
    rc = L.flame_delaunay_triangulate(p.ctypes.data_as(_FP), n, tri.ctypes.data_as(_IP), tri.shape[0], C.byref(nt),
                                      edg.ctypes.data_as(_IP), edg.shape[0], C.byref(ne))
    if rc != 0:
        raise NLTGV2Error(rc, "flame_delaunay_triangulate")
    return tri[: nt.value], edg[: ne.value]  # (views of the over-allocated arrays: no second copy)


def pack_probe(g: dict):
    """Host-only view of the SELL-64 layout the fused sweep uses (no GPU needed)."""
    L = load_library()
    keep = []
    cg = _graph_view(g, keep, need_all=False)
    rows = C.c_int64(0)
    n_slices = L.flame_nltgv2_pack_probe(C.byref(cg), None, None, None, None, 0, C.byref(rows))
    if n_slices < 0:
        raise NLTGV2Error(n_slices, "pack_probe")
    perm = np.empty(n_slices * 64, np.int32)
    slice_row = np.empty(n_slices + 1, np.int32)
    rec_nbr = np.empty(rows.value * 64, np.int32)
    rec_edge = np.empty(rows.value * 64, np.int32)
    rc = L.flame_nltgv2_pack_probe(C.byref(cg), perm.ctypes.data_as(_IP), slice_row.ctypes.data_as(_IP),
                                   rec_nbr.ctypes.data_as(_IP), rec_edge.ctypes.data_as(_IP), rows.value,
                                   C.byref(rows))
    if rc < 0:
        raise NLTGV2Error(rc, "pack_probe")
    return dict(n_slices=n_slices, rows=rows.value, perm=perm, slice_row=slice_row,
                rec_nbr=rec_nbr.view(np.uint32), rec_edge=rec_edge)


class Regularizer:
    """One solver context on one GPU; holds the device image of one Graph."""

    def __init__(self, device: int = 0):
        self._L = load_library()
        self._ctx = C.c_void_p()
        rc = self._L.flame_nltgv2_create(C.byref(self._ctx), int(device))
        if rc != 0:
            self._ctx = None
            raise NLTGV2Error(rc, "flame_nltgv2_create")
        self.V = self.E = 0

    def close(self):
        if getattr(self, "_ctx", None):
            self._L.flame_nltgv2_destroy(self._ctx)
            self._ctx = None

    __del__ = close

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def _chk(self, rc, what):
        if rc != 0:
            raise NLTGV2Error(rc, what)

    # ---- graph in / out -------------------------------------------------------------------------
    def upload_graph(self, g: dict):
        keep = []
        cg = _graph_view(g, keep)
        self._chk(self._L.flame_nltgv2_upload_graph(self._ctx, C.byref(cg)), "upload_graph")
        self.V, self.E = int(g["V"]), int(g["E"])

    def sync_prepare(self, *args, **kw):
        """First half of sync_graph (same arguments): checks, one staged copy, the new topology's construction enqueued on a side
        stream -- the solver may keep iterating (run_async) until sync_commit()."""
        self._sync(self._L.flame_nltgv2_sync_prepare, "sync_prepare", *args, **kw)

    def sync_commit(self):
        """Second half: waits for the construction, settles the runs, swaps the new topology in, moves the state."""
        self._chk(self._L.flame_nltgv2_sync_commit(self._ctx), "sync_commit")
        self._refresh_size()

    def _refresh_size(self):
        nv, ne = C.c_int32(0), C.c_int32(0)
        self._chk(self._L.flame_nltgv2_graph_size(self._ctx, C.byref(nv), C.byref(ne)), "graph_size")
        self.V, self.E = int(nv.value), int(ne.value)

    def sync_graph(self, *args, **kw):
        """Per-frame warm-start synchronisation (Flame::syncGraph's graph edits, flame.cc:1985-2121).

        init_graph_scale > 0: new vertices whose init_x is NaN start at their neighbours' mean
        (init_with_prediction's fallback, flame.cc:2133-2158)."""
        self._sync(self._L.flame_nltgv2_sync_graph, "sync_graph", *args, **kw)
        self._refresh_size()

    def _sync(self, fn, what, feat_id, pos, data_term, data_weight, edges, init_x=None, check_sticky_obstacles=False,
              sticky_threshold=0.25, init_graph_scale=0.0, edges_unique=False, init_from_map=False):
        V = int(len(feat_id))
        fid = _as(feat_id, np.int32, V, "feat_id")
        p = _as(pos, np.float32, 2 * V, "pos")
        d = _as(data_term, np.float32, V, "data_term")
        w = _as(data_weight, np.float32, V, "data_weight")
        ed = np.ascontiguousarray(edges, np.int32).reshape(-1, 2)
        si = _SyncInput()
        si.V, si.E = V, int(ed.shape[0])
        si.feat_id, si.pos = fid.ctypes.data_as(_IP), p.ctypes.data_as(_FP)
        si.data_term, si.data_weight = d.ctypes.data_as(_FP), w.ctypes.data_as(_FP)
        ix = None
        if init_x is not None:
            ix = _as(init_x, np.float32, V, "init_x")
            si.init_x = ix.ctypes.data_as(_FP)
        si.edges = ed.ctypes.data_as(_IP)
        si.check_sticky_obstacles = 1 if check_sticky_obstacles else 0
        si.sticky_threshold = sticky_threshold
        si.init_graph_scale = float(init_graph_scale)
        si.edges_unique = 1 if edges_unique else 0  # (the caller vouches: e.g. the edges of flame_amd.delaunay)
        si.init_from_map = 1 if init_from_map else 0  # new vertices start at the device-resident dense map's prediction (flame.cc:2131)
        self._chk(fn(self._ctx, C.byref(si)), what)

    def placement_info(self) -> dict:
        """Record placement of the patch-per-wave form (FLAME_NLTGV2_OPT_PLACEMENT): state (1 in use, 0 not yet, -1
        unavailable), records placed for the current topology, one-way hand-off (us) on the best / mean / worst page."""
        st, n = C.c_int32(0), C.c_int32(0)
        us = (C.c_float * 3)()
        self._chk(self._L.flame_nltgv2_placement_info(self._ctx, C.byref(st), C.byref(n), us), "placement_info")
        return {"state": int(st.value), "placed_records": int(n.value), "best_us": float(us[0]), "mean_us": float(us[1]),
                "worst_us": float(us[2])}

    def layout_selftest(self) -> int:
        """Words of the device-expanded layout arrays that differ from the host builders' (0 = identical)."""
        n = C.c_int64(-1)
        self._chk(self._L.flame_nltgv2_layout_selftest(self._ctx, C.byref(n)), "layout_selftest")
        return int(n.value)

    def set_feature_ids(self, feat_id):
        fid = _as(feat_id, np.int32, self.V, "feat_id")
        self._chk(self._L.flame_nltgv2_set_feature_ids(self._ctx, fid.ctypes.data_as(_IP)), "set_feature_ids")

    def topology(self):
        src = np.empty(self.E, np.int32)
        dst = np.empty(self.E, np.int32)
        fid = np.empty(self.V, np.int32)
        self._chk(self._L.flame_nltgv2_get_topology(self._ctx, src.ctypes.data_as(_IP), dst.ctypes.data_as(_IP),
                                                     fid.ctypes.data_as(_IP)), "get_topology")
        return src, dst, fid

    def project_graph(self, K, Kinv, KRKinv, q, t, region, graph_scale=1.0):
        """Flame::projectGraph on the device state; returns (keep mask, new positions)."""
        pr = _Projection()
        for name, a, n in (("K", K, 9), ("Kinv", Kinv, 9), ("KRKinv", KRKinv, 9), ("q_ref_to_cmp", q, 4), ("t_ref_to_cmp", t, 3)):
            arr = np.ascontiguousarray(a, np.float32).reshape(n)
            setattr(pr, name, (C.c_float * n)(*arr.tolist()))
        pr.region_x, pr.region_y, pr.region_w, pr.region_h = [float(r) for r in region]
        keep = np.zeros(self.V, np.uint8)
        pos = np.empty((self.V, 2), np.float32)
        self._chk(self._L.flame_nltgv2_project_graph(self._ctx, C.byref(pr), C.c_float(graph_scale),
                                                     keep.ctypes.data_as(C.POINTER(C.c_uint8)), pos.ctypes.data_as(_FP)),
                  "project_graph")
        return keep, pos

    def rescale_data(self, graph_scale, params: Params):
        """The rescale_data block (flame.cc:328-351); updates params.data_factor; returns the new graph scale."""
        ns = C.c_float(0)
        self._chk(self._L.flame_nltgv2_rescale_data(self._ctx, C.c_float(graph_scale), C.byref(ns), C.byref(params)),
                  "rescale_data")
        return float(ns.value)

    def update_data(self, data_term, data_weight):
        d = _as(data_term, np.float32, self.V, "data_term")
        w = _as(data_weight, np.float32, self.V, "data_weight")
        self._chk(self._L.flame_nltgv2_update_data(self._ctx, d.ctypes.data_as(_FP), w.ctypes.data_as(_FP)),
                  "update_data")

    def upload_state(self, state: dict):
        keep = []
        s = dict(state)
        s.setdefault("V", self.V)
        s.setdefault("E", self.E)
        cg = _graph_view(s, keep, need_all=False)
        self._chk(self._L.flame_nltgv2_upload_state(self._ctx, C.byref(cg)), "upload_state")

    def download_state(self, keys=VERTEX_STATE + EDGE_STATE) -> dict:
        out = {}
        cg = _Graph()
        for k in keys:
            n = self.V if k in VERTEX_STATE else self.E
            out[k] = np.empty(n, np.float32)
            setattr(cg, k, out[k].ctypes.data_as(_FP))
        self._chk(self._L.flame_nltgv2_download_state(self._ctx, C.byref(cg)), "download_state")
        return out

    # ---- the reference call surface -------------------------------------------------------------
    def run(self, params: Params, n_iters: int):
        """n_iters x step(params, graph), nltgv2...cc:33-49."""
        self._chk(self._L.flame_nltgv2_run(self._ctx, C.byref(params), int(n_iters)), "run")

    def step(self, params: Params):
        self._chk(self._L.flame_nltgv2_step(self._ctx, C.byref(params)), "step")

    def run_async(self, params: Params, n_iters: int):
        self._chk(self._L.flame_nltgv2_run_async(self._ctx, C.byref(params), int(n_iters)), "run_async")

    def sync(self):
        self._chk(self._L.flame_nltgv2_sync(self._ctx), "sync")

    def run_timed(self, params: Params, n_iters: int) -> float:
        """Returns device milliseconds (HIP events on the solver's stream)."""
        ms = C.c_float(0)
        self._chk(self._L.flame_nltgv2_run_timed(self._ctx, C.byref(params), int(n_iters), C.byref(ms)), "run_timed")
        return float(ms.value)

    def save_prev(self):
        self._chk(self._L.flame_nltgv2_save_prev(self._ctx), "save_prev")

    def dual_step(self, params: Params):
        self._chk(self._L.flame_nltgv2_dual_step(self._ctx, C.byref(params)), "dual_step")

    def primal_step(self, params: Params):
        self._chk(self._L.flame_nltgv2_primal_step(self._ctx, C.byref(params)), "primal_step")

    def extragradient_step(self, params: Params):
        self._chk(self._L.flame_nltgv2_extragradient_step(self._ctx, C.byref(params)), "extragradient_step")

    def costs(self, params: Params):
        s, d = C.c_float(0), C.c_float(0)
        self._chk(self._L.flame_nltgv2_costs(self._ctx, C.byref(params), C.byref(s), C.byref(d)), "costs")
        return float(s.value), float(d.value)

    def smoothness_cost(self, params: Params) -> float:
        return self.costs(params)[0]

    def data_cost(self, params: Params) -> float:
        return self.costs(params)[1]

    def cost(self, params: Params) -> float:
        s, d = self.costs(params)
        return float(np.float32(s) + np.float32(d))

    # ---- mesh -> dense inverse-depth map (utils::interpolateMesh) -----------------------------------
    def interpolate_mesh(self, triangles, rows, cols, graph_scale=1.0, tri_valid=None):
        """Rasterises the solver's current x*graph_scale over `triangles`; returns (idepthmap, coverage)."""
        tr = np.ascontiguousarray(triangles, np.int32).reshape(-1, 3)
        img = np.empty((rows, cols), np.float32)
        cov = C.c_int32(0)
        U8 = C.POINTER(C.c_uint8)
        tv = None if tri_valid is None else np.ascontiguousarray(tri_valid, np.uint8)
        self._chk(self._L.flame_nltgv2_interpolate_mesh(self._ctx, tr.ctypes.data_as(_IP), tr.shape[0],
                                                        None if tv is None else tv.ctypes.data_as(U8), rows, cols,
                                                        C.c_float(graph_scale), img.ctypes.data_as(_FP), C.byref(cov)),
                  "interpolate_mesh")
        return img, int(cov.value)

    def interpolate_mesh_begin(self, triangles, rows, cols, graph_scale=1.0, tri_valid=None):
        """interpolate_mesh in two halves: settles the runs, enqueues rasteriser + copy-out on a side stream and returns; the
        solver may iterate (run_async) until interpolate_mesh_end() fetches the map."""
        tr = np.ascontiguousarray(triangles, np.int32).reshape(-1, 3)
        U8 = C.POINTER(C.c_uint8)
        tv = None if tri_valid is None else np.ascontiguousarray(tri_valid, np.uint8)
        self._chk(self._L.flame_nltgv2_interpolate_mesh_begin(self._ctx, tr.ctypes.data_as(_IP), tr.shape[0],
                                                              None if tv is None else tv.ctypes.data_as(U8), rows, cols,
                                                              C.c_float(graph_scale)), "interpolate_mesh_begin")
        self._map_shape = (rows, cols)

    def interpolate_mesh_end(self, copy=True):
        """-> (idepthmap, coverage); copy=False: a view of the context's pinned buffer, valid until the next begin."""
        p, cov = _FP(), C.c_int32(0)
        self._chk(self._L.flame_nltgv2_interpolate_mesh_end(self._ctx, C.byref(p), C.byref(cov)), "interpolate_mesh_end")
        rows, cols = self._map_shape
        img = np.ctypeslib.as_array(p, shape=(rows, cols))
        return (img.copy() if copy else img), int(cov.value)

    def interpolate_mesh_arrays(self, triangles, vertices, values, rows, cols, vtx_valid=None, tri_valid=None):
        tr = np.ascontiguousarray(triangles, np.int32).reshape(-1, 3)
        xy = np.ascontiguousarray(vertices, np.float32).reshape(-1, 2)
        val = _as(values, np.float32, xy.shape[0], "values")
        img = np.empty((rows, cols), np.float32)
        cov = C.c_int32(0)
        U8 = C.POINTER(C.c_uint8)
        tv = None if tri_valid is None else np.ascontiguousarray(tri_valid, np.uint8)
        vv = None if vtx_valid is None else np.ascontiguousarray(vtx_valid, np.uint8)
        self._chk(self._L.flame_nltgv2_interpolate_mesh_arrays(
            self._ctx, tr.ctypes.data_as(_IP), tr.shape[0], xy.ctypes.data_as(_FP), val.ctypes.data_as(_FP), xy.shape[0],
            None if vv is None else vv.ctypes.data_as(U8), None if tv is None else tv.ctypes.data_as(U8), rows, cols,
            img.ctypes.data_as(_FP), C.byref(cov)), "interpolate_mesh_arrays")
        return img, int(cov.value)

    # ---- config-5 epilogue ----------------------------------------------------------------------
    def photo_set_images(self, ref, cmp):
        ref = np.ascontiguousarray(ref, np.uint8)
        cmp = np.ascontiguousarray(cmp, np.uint8)
        if ref.shape != cmp.shape or ref.ndim != 2:
            raise ValueError("two equally sized single-channel u8 images expected")
        U8 = C.POINTER(C.c_uint8)
        self._chk(self._L.flame_nltgv2_photo_set_images(self._ctx, ref.ctypes.data_as(U8), cmp.ctypes.data_as(U8),
                                                         ref.shape[0], ref.shape[1], ref.strides[0]), "photo_set_images")

    def photo_residual(self, KRKinv, Kt, graph_scale=1.0, border=3):
        k = np.ascontiguousarray(KRKinv, np.float32).reshape(9)
        t = np.ascontiguousarray(Kt, np.float32).reshape(3)
        err = np.empty(self.V, np.float32)
        self._chk(self._L.flame_nltgv2_photo_residual(self._ctx, k.ctypes.data_as(_FP), t.ctypes.data_as(_FP),
                                                       C.c_float(graph_scale), int(border), err.ctypes.data_as(_FP)),
                  "photo_residual")
        return err

    # ---- plumbing -------------------------------------------------------------------------------
    def photo_fuse(self, KRKinv=None, Kt=None, graph_scale=1.0, border=3, enable=True):
        """While enabled, every run also leaves the photometric residual of its final x on the device (config 5)."""
        if not enable:
            self._chk(self._L.flame_nltgv2_photo_fuse(self._ctx, None, None, C.c_float(1.0), 3, 0), "photo_fuse")
            return
        k = np.ascontiguousarray(KRKinv, np.float32).reshape(9)
        t = np.ascontiguousarray(Kt, np.float32).reshape(3)
        self._chk(self._L.flame_nltgv2_photo_fuse(self._ctx, k.ctypes.data_as(_FP), t.ctypes.data_as(_FP),
                                                  C.c_float(graph_scale), int(border), 1), "photo_fuse")

    def photo_residual_last(self) -> np.ndarray:
        err = np.empty(self.V, np.float32)
        self._chk(self._L.flame_nltgv2_photo_residual_last(self._ctx, err.ctypes.data_as(_FP)), "photo_residual_last")
        return err

    def set_stream(self, hip_stream_ptr: int | None):
        self._chk(self._L.flame_nltgv2_set_stream(self._ctx, C.c_void_p(hip_stream_ptr or 0)), "set_stream")

    def runs_in_flight(self) -> int:
        """How many of the last two run_async() runs the device has not finished yet (non-blocking)."""
        n = C.c_int32(0)
        self._chk(self._L.flame_nltgv2_runs_in_flight(self._ctx, C.byref(n)), "runs_in_flight")
        return int(n.value)

    def run_open(self, params: Params, max_iters: int) -> bool:
        """A run that goes on (at most max_iters, even) until the next call that needs the solver settled asks it to stop.  False: not
        applicable to this graph / configuration -- nothing was enqueued, use run_async."""
        opened = C.c_int32(0)
        self._chk(self._L.flame_nltgv2_run_open(self._ctx, C.byref(params), int(max_iters), C.byref(opened)), "run_open")
        return bool(opened.value)

    def iterations(self):
        """(iterations applied by all runs so far, whether an open run is in flight and not yet counted); never waits."""
        total, open_ = C.c_int64(0), C.c_int32(0)
        self._chk(self._L.flame_nltgv2_iterations(self._ctx, C.byref(total), C.byref(open_)), "iterations")
        return int(total.value), bool(open_.value)

    def stream_wait_run(self, hip_stream_ptr: int):
        """Another stream of the caller waits for everything enqueued on the solver's stream so far (right behind run_async: at no cost to
        the solver's stream -- the launch carries the event)."""
        self._chk(self._L.flame_nltgv2_stream_wait_run(self._ctx, C.c_void_p(hip_stream_ptr)), "stream_wait_run")

    def set_option(self, option: int, value: int):
        self._chk(self._L.flame_nltgv2_set_option(self._ctx, int(option), int(value)), "set_option")

    def export_idepth_device(self, device_ptr: int, scale: float = 1.0, wait: bool = True):
        fn = self._L.flame_nltgv2_export_idepth_device if wait else self._L.flame_nltgv2_export_idepth_device_async
        self._chk(fn(self._ctx, C.c_void_p(device_ptr), C.c_float(scale)), "export_idepth_device")

    def set_export_target(self, device_ptr: int | None, scale: float = 1.0):
        """While set, every run also leaves scale * x (original vertex order) in this device buffer."""
        self._chk(self._L.flame_nltgv2_set_export_target(self._ctx, C.c_void_p(device_ptr or 0), C.c_float(scale)),
                  "set_export_target")

    def read_probe(self) -> np.ndarray:
        """Cycle probe of the last patch-per-wave run (OPT_PROBE): uint32 [patches, steps, 8] flattened."""
        n = C.c_int64(0)
        self._chk(self._L.flame_nltgv2_read_probe(self._ctx, None, 0, C.byref(n)), "read_probe")
        out = np.zeros(int(n.value), dtype=np.uint32)
        if out.size:
            self._chk(self._L.flame_nltgv2_read_probe(self._ctx, out.ctypes.data_as(C.POINTER(C.c_uint32)), out.size,
                                                      C.byref(n)), "read_probe")
        return out

    def info(self) -> dict:
        i = _Info()
        self._chk(self._L.flame_nltgv2_get_info(self._ctx, C.byref(i)), "get_info")
        d = {n: getattr(i, n) for n, _ in _Info._fields_}
        d["device_name"] = d["device_name"].decode(errors="replace")
        d["gcn_arch"] = d["gcn_arch"].decode(errors="replace")
        return d

    def last_error(self) -> int:
        return int(self._L.flame_nltgv2_last_error(self._ctx))
