"""Generates tests/golden/*.npz (run in the build container: needs oracle/_ref/libref_triangle.so,
i.e. /root/reference).  Usage:  python -m oracle.make_golden

What the fixtures are -- and are not:
  * INPUTS: synthetic vertices (flame_amd.synth) triangulated by the REFERENCE's vendored Triangle
    with the reference's switches, so edge order and (source,target) orientation are exactly what
    flame.cc:2085-2096 would feed boost::add_edge.
  * OUTPUTS: solver state after N steps computed by oracle/nltgv2_oracle.c, cross-checked bit-exact
    against oracle/nltgv2_numpy.py.  They are NOT outputs of the reference binary (unbuildable here,
    see nltgv2_oracle.c header) -> PARITY UNPINNED; they freeze the restatement so that the on-box
    GPU run is checked against numbers that were produced and reviewed in the build container.
"""
from __future__ import annotations

import os

import numpy as np

from flame_amd import synth
from oracle import capi, nltgv2_numpy, ref_triangle

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
INPUT_KEYS = ("pos", "data_term", "data_weight", "src", "dst", "alpha", "beta")


def make(config: str, seed: int, iters, full_at, weights: str = "ones"):
    g = synth.make_graph(config, seed, delaunay=ref_triangle.delaunay_edges)
    if weights == "varied":  # adaptive_data_weights-like (flame.cc:2043-2044) + non-unit beta
        g["data_weight"] = (np.float32(0.25) + np.float32(3.0) * synth.uniform01(seed, g["V"], stream=9)).astype(np.float32)
        g["beta"] = (np.float32(0.5) + synth.uniform01(seed, g["E"], stream=10)).astype(np.float32)
    out = {k: g[k] for k in INPUT_KEYS}
    out["config"] = np.array(config)
    out["seed"] = np.array(seed)
    out["iters"] = np.array(sorted(iters))
    a = synth.copy_graph(g)
    b = synth.copy_graph(g)
    done = 0
    for n in sorted(iters):
        capi.run(a, n - done)
        nltgv2_numpy.run(b, n - done, capi.DEFAULT_PARAMS)
        done = n
        for k in synth.STATE_KEYS:
            assert np.array_equal(a[k], b[k]), (config, n, k)
        keys = synth.STATE_KEYS if n in full_at else ("x", "w1", "w2")
        for k in keys:
            if k.endswith("_prev"):
                continue
            out[f"n{n}_{k}"] = a[k].copy()
        sm, dc = capi.costs(a)
        out[f"n{n}_cost"] = np.array([sm, dc], dtype=np.float32)
    return out


def make_delaunay_fixture():
    """Point sets + the triangulations the REFERENCE's vendored Triangle produces for them (oracle/_ref):
    reference-run outputs that pin flame_delaunay_triangulate (tests/test_delaunay.py)."""
    rng = np.random.default_rng(2024)
    sets = {
        "jittered_320x240": synth.make_points(320, 240, 6, 99),
        "uniform_1500": (rng.random((1500, 2)) * [640, 480]).astype(np.float32),
        "clustered_900": np.concatenate([rng.normal(c, 9.0, (300, 2)) for c in ((100, 100), (300, 200), (500, 380))]).astype(np.float32),
        "tiny_7": (rng.random((7, 2)) * 50).astype(np.float32),
    }
    out = {}
    for name, pts in sets.items():
        out[name + "_points"] = pts
        out[name + "_triangles"] = ref_triangle.delaunay_triangles(pts)
        out[name + "_edges"] = ref_triangle.delaunay_edges(pts)
    path = os.path.join(OUT, "delaunay_ref_triangle.npz")
    np.savez_compressed(path, **out)
    print("delaunay_ref_triangle", {k: v.shape for k, v in out.items() if k.endswith("triangles")}, os.path.getsize(path) // 1024, "KiB")


def canonical_digests(tris, edges):
    """SHA-256 of the sorted triangle set (each triangle rotated to start at its smallest vertex: the winding is kept) and of the sorted
    undirected edge set."""
    import hashlib

    t = np.asarray(tris, dtype=np.int64).reshape(-1, 3)
    k = np.argmin(t, axis=1)
    t = np.take_along_axis(t, (np.arange(3)[None, :] + k[:, None]) % 3, 1)
    t = t[np.lexsort((t[:, 2], t[:, 1], t[:, 0]))]
    e = np.sort(np.asarray(edges, dtype=np.int64).reshape(-1, 2), axis=1)
    e = e[np.lexsort((e[:, 1], e[:, 0]))]
    return hashlib.sha256(t.astype("<i4").tobytes()).hexdigest(), hashlib.sha256(e.astype("<i4").tobytes()).hexdigest()


def large_delaunay_sets():
    """Point sets of more than 4096 points -- what flame_delaunay_triangulate cuts into strips -- regenerated from their seeds."""
    rng = np.random.default_rng(4096)
    return {
        "jittered_640x480": synth.make_points(640, 480, 6, 7),
        "uniform_6000": (rng.random((6000, 2)) * [640, 480]).astype(np.float32),
        "clustered_5400": np.concatenate([rng.normal(c, s, (1800, 2)) for c, s in (((120, 100), 25.0), ((330, 260), 60.0), ((520, 380), 12.0))]).astype(np.float32),
    }


def make_large_delaunay_fixture():
    """Digests of what the REFERENCE's vendored Triangle (oracle/_ref) makes of the large sets: they pin the strip paths of the
    triangulator (merged strips, certified strips) the way delaunay_ref_triangle.npz pins the sequential one."""
    import json

    out = {}
    for name, pts in large_delaunay_sets().items():
        t, e = ref_triangle.delaunay_triangles(pts), ref_triangle.delaunay_edges(pts)
        ht, he = canonical_digests(t, e)
        out[name] = {"points": int(len(pts)), "triangles": int(len(t)), "edges": int(len(e)), "triangles_sha256": ht, "edges_sha256": he}
    path = os.path.join(OUT, "delaunay_ref_triangle_large.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1)
    print("delaunay_ref_triangle_large", {k: (v["points"], v["triangles"]) for k, v in out.items()})


def main():
    os.makedirs(OUT, exist_ok=True)
    make_large_delaunay_fixture()
    if os.environ.get("GOLDEN_ONLY_DELAUNAY_LARGE"):
        return
    make_delaunay_fixture()
    if os.environ.get("GOLDEN_ONLY_DELAUNAY"):
        return
    jobs = [
        ("cfg1_320x240_s1234", dict(config="320x240", seed=1234, iters=(1, 2, 50, 200), full_at=(1, 2, 50, 200))),
        ("cfg1_320x240_s77_varied", dict(config="320x240", seed=77, iters=(1, 50), full_at=(50,), weights="varied")),
        ("cfg2_640x480_s1234", dict(config="640x480", seed=1234, iters=(1, 50, 200), full_at=(200,))),
    ]
    for name, kw in jobs:
        d = make(**kw)
        path = os.path.join(OUT, name + ".npz")
        np.savez_compressed(path, **d)
        print(name, "V", d["pos"].shape[0], "E", d["src"].shape[0], os.path.getsize(path) // 1024, "KiB")


if __name__ == "__main__":
    main()
