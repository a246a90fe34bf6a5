"""Synthetic solver inputs (SURVEY.md section 8(d) "Synthetic inputs").

Produces the graphs BASELINE.json's configs name: a jittered grid of feature points on a WxH image
(one point per c x c px cell), Delaunay-triangulated, with a two-plane inverse-depth scene plus
noise and outliers as the data term.  Vertex/edge initialisation follows the conventions of the
reference's graph builder:

  * new vertex: x = x_bar = x_prev = data_term, w = 0, data_weight = 1
    (/root/reference/src/flame/flame.cc:2040-2048, params.h:91)
  * new edge:   alpha = 1/||pos_i - pos_j||, beta = 1, q = 0
    (flame.cc:2087-2104, nltgv2_l1_graph_regularizer.h:96-100)

The RNG is a counter-based splitmix64 defined here so inputs are reproducible across hosts.
Delaunay on this side is the library's own triangulator (flame_delaunay_triangulate; scipy/Qhull variants
are kept for cross-checks); fixtures under tests/golden/ instead carry edge lists from the reference's own
vendored Triangle (see oracle/make_golden.py).  The solver takes explicit edge lists, so any source is
valid input.
"""
from __future__ import annotations

import numpy as np

_GAMMA = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)

# BASELINE.json configs -> (W, H, cell px).  SURVEY.md section 8: cfg1..cfg5.
CONFIGS = {
    "320x240": (320, 240, 6),
    "640x480": (640, 480, 6),
    "1280x720": (1280, 720, 7),
    "1920x1080": (1920, 1080, 6),
}


def splitmix64(seed: int, n: int, stream: int = 0) -> np.ndarray:
    """n 64-bit outputs of splitmix64 started at `seed` (+ a per-stream offset)."""
    with np.errstate(over="ignore"):
        base = np.uint64(seed & 0xFFFFFFFFFFFFFFFF) + np.uint64(stream) * np.uint64(0xD1B54A32D192ED03)
        z = base + (np.arange(1, n + 1, dtype=np.uint64) * _GAMMA)
        z = (z ^ (z >> np.uint64(30))) * _M1
        z = (z ^ (z >> np.uint64(27))) * _M2
        z = z ^ (z >> np.uint64(31))
    return z


def uniform01(seed: int, n: int, stream: int = 0) -> np.ndarray:
    """float32 uniforms in [0,1) with 24 random bits (exactly representable)."""
    z = splitmix64(seed, n, stream)
    return ((z >> np.uint64(40)).astype(np.float32)) * np.float32(2.0**-24)


def make_points(width: int, height: int, cell: int, seed: int) -> np.ndarray:
    nx, ny = width // cell, height // cell
    n = nx * ny
    u = uniform01(seed, n, stream=1)
    v = uniform01(seed, n, stream=2)
    cx = np.tile(np.arange(nx, dtype=np.float32), ny)
    cy = np.repeat(np.arange(ny, dtype=np.float32), nx)
    pos = np.empty((n, 2), dtype=np.float32)
    pos[:, 0] = cx * np.float32(cell) + u * np.float32(cell - 1)
    pos[:, 1] = cy * np.float32(cell) + v * np.float32(cell - 1)
    return pos


def make_data_term(pos: np.ndarray, width: int, height: int, seed: int) -> np.ndarray:
    n = pos.shape[0]
    xh = pos[:, 0].astype(np.float64) / width
    yh = pos[:, 1].astype(np.float64) / height
    plane = np.where(xh < 0.5, 0.5 + 0.8 * xh + 0.2 * yh, 1.6 - 0.5 * xh + 0.3 * yh)
    u1 = uniform01(seed, n, stream=3).astype(np.float64)
    u2 = uniform01(seed, n, stream=4).astype(np.float64)
    gauss = np.sqrt(-2.0 * np.log(np.maximum(u1, 2.0**-24))) * np.cos(2.0 * np.pi * u2)
    d = plane + 0.05 * gauss
    is_out = uniform01(seed, n, stream=5) < np.float32(0.05)
    outl = 2.0 * uniform01(seed, n, stream=6).astype(np.float64)
    d = np.where(is_out, outl, d)
    # round through 1e-6 steps so that libm last-bit differences between hosts cannot change the
    # float32 value
    d = np.round(np.maximum(d, 0.0) * 1e6) / 1e6
    return d.astype(np.float32)


def delaunay_edges_scipy(pos: np.ndarray) -> np.ndarray:
    """Unique undirected Delaunay edges as an (E,2) int32 array.  Orientation = the direction in
    which the edge is first met walking the simplices (a,b),(b,c),(c,a); order = first occurrence."""
    from scipy.spatial import Delaunay

    tri = Delaunay(pos.astype(np.float64)).simplices.astype(np.int64)
    cand = np.concatenate([tri[:, [0, 1]], tri[:, [1, 2]], tri[:, [2, 0]]], axis=1).reshape(-1, 2)
    lo = np.minimum(cand[:, 0], cand[:, 1])
    hi = np.maximum(cand[:, 0], cand[:, 1])
    key = lo * np.int64(pos.shape[0]) + hi
    _, first = np.unique(key, return_index=True)
    first.sort()
    return cand[first].astype(np.int32)


def delaunay_native(pos: np.ndarray):
    """(triangles, edges) from the library's own triangulator (flame_delaunay_triangulate: host code, exact
    predicates).  The ORDER of the edges depends on how many strips the triangulator cuts the input into; the synthetic
    graphs (and the committed fixtures of them, tests/golden/config_hashes.json) are defined with the strip count of rounds
    1-3, min(32, n / 1024) from 4096 points, certified strips -- pinned here, whatever the library's default has become since."""
    import os

    from .regularizer import delaunay

    n = int(np.asarray(pos).reshape(-1, 2).shape[0])
    legacy = 1 if n < 4096 else min(32, n // 1024)
    pins = {"FLAME_DELAUNAY_STRIPS": str(legacy), "FLAME_DELAUNAY_MERGE": "0"}  # (round 5's merged strips order the output differently too)
    before = {k: os.environ.get(k) for k in pins}
    os.environ.update(pins)
    try:
        return delaunay(pos)
    finally:
        for k, v in before.items():
            if v is None:
                del os.environ[k]
            else:
                os.environ[k] = v


def delaunay_edges_native(pos: np.ndarray) -> np.ndarray:
    return delaunay_native(pos)[1]


def delaunay_triangles_scipy(pos: np.ndarray) -> np.ndarray:
    """Delaunay triangles as (T,3) int32 with the winding of the reference's triangulator (Shewchuk
    Triangle: counter-clockwise in x-right / y-up coordinates, i.e. positive signed area)."""
    from scipy.spatial import Delaunay

    tri = Delaunay(pos.astype(np.float64)).simplices.astype(np.int32)
    p = pos.astype(np.float64)
    a, b, c = p[tri[:, 0]], p[tri[:, 1]], p[tri[:, 2]]
    area2 = (b[:, 0] - a[:, 0]) * (c[:, 1] - a[:, 1]) - (c[:, 0] - a[:, 0]) * (b[:, 1] - a[:, 1])
    flip = area2 < 0
    tri[flip] = tri[flip][:, [0, 2, 1]]
    return np.ascontiguousarray(tri)


def edge_weights(pos: np.ndarray, edges: np.ndarray):
    """alpha = 1/||pos_i - pos_j|| in float32 (flame.cc:2087-2102), beta = 1 (flame.cc:2103)."""
    d = pos[edges[:, 0]] - pos[edges[:, 1]]
    length = np.sqrt(d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1], dtype=np.float32)
    alpha = (np.float32(1.0) / length).astype(np.float32)
    beta = np.ones(edges.shape[0], dtype=np.float32)
    return alpha, beta


def assemble_graph(pos: np.ndarray, data: np.ndarray, edges: np.ndarray, weight=None) -> dict:
    """Graph dict in the flat layout of include/flame_nltgv2.h with freshly initialised state."""
    V, E = pos.shape[0], edges.shape[0]
    alpha, beta = edge_weights(pos, edges) if E else (np.zeros(0, np.float32), np.zeros(0, np.float32))
    z = lambda n: np.zeros(n, dtype=np.float32)  # noqa: E731
    g = dict(
        V=V, E=E,
        pos=np.ascontiguousarray(pos, dtype=np.float32),
        data_term=np.ascontiguousarray(data, dtype=np.float32),
        data_weight=np.ones(V, dtype=np.float32) if weight is None else np.ascontiguousarray(weight, np.float32),
        src=np.ascontiguousarray(edges[:, 0], dtype=np.int32) if E else np.zeros(0, np.int32),
        dst=np.ascontiguousarray(edges[:, 1], dtype=np.int32) if E else np.zeros(0, np.int32),
        alpha=alpha, beta=beta,
        x=data.astype(np.float32).copy(), w1=z(V), w2=z(V),
        x_bar=data.astype(np.float32).copy(), w1_bar=z(V), w2_bar=z(V),
        x_prev=data.astype(np.float32).copy(), w1_prev=z(V), w2_prev=z(V),
        q1=z(E), q2=z(E), q3=z(E),
    )
    return g


def make_graph(config: str = "640x480", seed: int = 1234, delaunay=delaunay_edges_native) -> dict:
    width, height, cell = CONFIGS[config]
    pos = make_points(width, height, cell, seed)
    data = make_data_term(pos, width, height, seed)
    edges = delaunay(pos)
    g = assemble_graph(pos, data, edges)
    g["config"] = config
    g["seed"] = seed
    return g


STATE_KEYS = ("x", "w1", "w2", "x_bar", "w1_bar", "w2_bar", "x_prev", "w1_prev", "w2_prev", "q1", "q2", "q3")


def copy_graph(g: dict) -> dict:
    return {k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in g.items()}


def concat_graphs(graphs) -> dict:
    """Disjoint union of graphs (vertex ids offset) -- how a batch of independent frames is handed
    to one solver context."""
    out = {}
    voff = 0
    srcs, dsts = [], []
    for g in graphs:
        srcs.append(g["src"] + np.int32(voff))
        dsts.append(g["dst"] + np.int32(voff))
        voff += g["V"]
    for k in ("pos", "data_term", "data_weight", "alpha", "beta") + STATE_KEYS:
        out[k] = np.ascontiguousarray(np.concatenate([g[k] for g in graphs], axis=0))
    out["src"] = np.ascontiguousarray(np.concatenate(srcs)).astype(np.int32)
    out["dst"] = np.ascontiguousarray(np.concatenate(dsts)).astype(np.int32)
    out["V"] = voff
    out["E"] = int(out["src"].shape[0])
    return out
