"""flame_delaunay_triangulate (host code, like the reference's utils::Delaunay).

PINNED by reference-run outputs: tests/golden/delaunay_ref_triangle.npz holds point sets together with the
triangulations the reference's own vendored Triangle ("zneQB", delaunay.cc:66-68) produced for them in the
build container (oracle/make_golden.py, oracle/_ref).  A point set in general position has exactly one
Delaunay triangulation, so the triangle SETS must be equal; output order / rotation are the implementation's
own.  Also: agreement with Qhull (scipy) at every BASELINE size, exact empty-circumcircle verification with
Python integers on degenerate (co-circular) input, and the degenerate / error cases."""
import os
from fractions import Fraction

import numpy as np
import pytest

from flame_amd import regularizer, synth
from tests.conftest import ROOT
from tests.helpers import GOLDEN


def tri_set(t):
    return set(tuple(sorted(int(v) for v in x)) for x in np.asarray(t).reshape(-1, 3))


def edge_set(e):
    return set(tuple(sorted(int(v) for v in x)) for x in np.asarray(e).reshape(-1, 2))


def signed_area2(p, t):
    a, b, c = p[t[:, 0]].astype(np.float64), p[t[:, 1]].astype(np.float64), p[t[:, 2]].astype(np.float64)
    return (b[:, 0] - a[:, 0]) * (c[:, 1] - a[:, 1]) - (c[:, 0] - a[:, 0]) * (b[:, 1] - a[:, 1])


@pytest.mark.parametrize("name", ["jittered_320x240", "uniform_1500", "clustered_900", "tiny_7"])
def test_equals_reference_triangle_run(built, name):
    z = np.load(os.path.join(GOLDEN, "delaunay_ref_triangle.npz"))
    pts = z[name + "_points"]
    tri, edg = regularizer.delaunay(pts)
    assert tri_set(tri) == tri_set(z[name + "_triangles"])
    assert edge_set(edg) == edge_set(z[name + "_edges"])
    assert len(edg) == len(edge_set(edg))  # unique
    assert np.all(signed_area2(pts, tri) > 0)  # counter-clockwise, Triangle's convention
    assert np.all(signed_area2(pts, z[name + "_triangles"]) > 0)


@pytest.mark.parametrize("config", ["320x240", "640x480", "1280x720", "1920x1080"])
def test_equals_qhull_at_baseline_sizes(built, config):
    from scipy.spatial import Delaunay

    w, h, c = synth.CONFIGS[config]
    pts = synth.make_points(w, h, c, 31)
    tri, edg = regularizer.delaunay(pts)
    assert tri_set(tri) == tri_set(Delaunay(pts.astype(np.float64)).simplices)
    assert edge_set(edg) == edge_set(synth.delaunay_edges_scipy(pts))
    n = len(pts)
    assert len(edg) == len(tri) + n - 1  # Euler: V - E + T = 1 for a triangulated disk


def exact_incircle(pa, pb, pc, pd):
    f = lambda v: Fraction(float(v))  # noqa: E731
    adx, ady = f(pa[0]) - f(pd[0]), f(pa[1]) - f(pd[1])
    bdx, bdy = f(pb[0]) - f(pd[0]), f(pb[1]) - f(pd[1])
    cdx, cdy = f(pc[0]) - f(pd[0]), f(pc[1]) - f(pd[1])
    return ((adx * adx + ady * ady) * (bdx * cdy - cdx * bdy) + (bdx * bdx + bdy * bdy) * (cdx * ady - adx * cdy)
            + (cdx * cdx + cdy * cdy) * (adx * bdy - bdx * ady))


def test_cocircular_grid_is_a_valid_delaunay_triangulation(built):
    """An exact grid is maximally degenerate (every cell is co-circular): the triangulation is not unique,
    so it is verified instead -- exact empty-circumcircle test of every triangle against the opposite vertex
    of every neighbour (rational arithmetic), full coverage, consistent winding."""
    xs, ys = np.meshgrid(np.arange(0, 13, dtype=np.float32) * 7.5, np.arange(0, 9, dtype=np.float32) * 7.5)
    pts = np.stack([xs.ravel(), ys.ravel()], 1)
    tri, edg = regularizer.delaunay(pts)
    n = len(pts)
    hull = 2 * (13 + 9) - 4
    assert len(tri) == 2 * n - 2 - hull and len(edg) == 3 * n - 3 - hull
    assert np.all(signed_area2(pts, tri) > 0)
    assert abs(0.5 * signed_area2(pts, tri).sum() - (12 * 7.5) * (8 * 7.5)) < 1e-6
    by_edge = {}
    for t in tri:
        for i in range(3):
            by_edge.setdefault(tuple(sorted((int(t[i]), int(t[(i + 1) % 3])))), []).append(t)
    for (u, v), ts in by_edge.items():
        assert len(ts) <= 2
        if len(ts) == 2:
            for a, b in ((ts[0], ts[1]), (ts[1], ts[0])):
                opp = [int(w) for w in b if int(w) not in (u, v)][0]
                assert exact_incircle(pts[a[0]], pts[a[1]], pts[a[2]], pts[opp]) <= 0  # not strictly inside


def test_points_on_edges_and_hull(built):
    """Points exactly ON an existing edge and exactly on the hull line (collinear) are inserted correctly."""
    pts = np.array([[0, 0], [10, 0], [0, 10], [10, 10], [5, 5], [5, 0], [2.5, 2.5], [20, 0], [7.5, 7.5]], np.float32)
    tri, edg = regularizer.delaunay(pts)
    assert np.all(signed_area2(pts, tri) > 0)
    assert abs(0.5 * signed_area2(pts, tri).sum() - 150.0) < 1e-9  # area of the hull (0,0),(20,0),(10,10),(0,10)
    assert set(np.unique(tri)) == set(range(len(pts)))
    from scipy.spatial import Delaunay

    assert len(tri) == len(Delaunay(pts.astype(np.float64)).simplices)


def test_degenerate_and_invalid_inputs(built):
    for pts in (np.zeros((0, 2), np.float32), np.array([[1, 1]], np.float32), np.array([[0, 0], [1, 1]], np.float32),
                np.array([[0, 0], [1, 1], [2, 2], [3, 3]], np.float32),  # collinear
                np.array([[3, 4]] * 5, np.float32)):  # all identical
        tri, edg = regularizer.delaunay(pts)
        assert len(tri) == 0 and len(edg) == 0
    pts = np.array([[0, 0], [4, 0], [0, 4], [4, 0], [0, 0], [1, 1]], np.float32)  # exact duplicates are skipped
    tri, edg = regularizer.delaunay(pts)
    assert tri_set(tri) == {(0, 1, 5), (0, 2, 5), (1, 2, 5)}
    bad = np.array([[0, 0], [1, 0], [np.nan, 1]], np.float32)
    with pytest.raises(regularizer.NLTGV2Error):
        regularizer.delaunay(bad)
    huge = np.array([[0, 0], [1e-30, 0], [1e30, 1]], np.float32)  # dynamic range beyond the exact predicates
    with pytest.raises(regularizer.NLTGV2Error):
        regularizer.delaunay(huge)


def test_result_feeds_the_solver_graph(built):
    """The triangulator's edges are a valid solver topology (what syncGraph does with them, flame.cc:2085-2104)."""
    pts = synth.make_points(320, 240, 6, 5)
    _, edg = regularizer.delaunay(pts)
    g = synth.assemble_graph(pts, synth.make_data_term(pts, 320, 240, 5), edg)
    pr = regularizer.pack_probe(g)
    assert pr["n_slices"] == (g["V"] + 63) // 64


def _sets(tris, edges):
    return ({tuple(sorted(map(int, t))) for t in tris}, {tuple(sorted(map(int, e))) for e in edges})


def _triangulate_in_subprocess(pos, strips, threads):
    """flame_delaunay_triangulate in a fresh process (the worker pool and the strip count are read from the environment once)."""
    import subprocess
    import sys
    import tempfile

    with tempfile.TemporaryDirectory() as d:
        np.save(os.path.join(d, "pos.npy"), pos)
        code = ("import sys, numpy as np; sys.path.insert(0, %r); from flame_amd.regularizer import delaunay; "
                "t, e = delaunay(np.load(%r)); np.save(%r, t); np.save(%r, e)"
                % (ROOT, os.path.join(d, "pos.npy"), os.path.join(d, "t.npy"), os.path.join(d, "e.npy")))
        env = dict(os.environ, FLAME_DELAUNAY_THREADS=str(threads))
        if strips:
            env["FLAME_DELAUNAY_STRIPS"] = str(strips)
        subprocess.check_call([sys.executable, "-c", code], env=env)
        return np.load(os.path.join(d, "t.npy")), np.load(os.path.join(d, "e.npy"))


@pytest.mark.parametrize("case", ["jittered grid", "exact grid (all co-circular)", "clustered + sparse", "duplicates + collinear rows"])
def test_parallel_strips_equal_the_sequential_triangulation(built, case):
    """Round 3: inputs of >= 4096 points are triangulated as certified strips on worker threads.  Same triangle / edge SETS as
    the sequential build (one strip), for any thread count; identical OUTPUT (order included) for 1 and 4 threads -- the
    number of strips depends on the input only; the cases that cannot be certified (sparse regions, co-circular grids whose
    ties the strips would break differently) fall back to the sequential build and still agree."""
    rng = np.random.default_rng(5)
    if case == "jittered grid":
        pos = synth.make_points(640, 480, 6, 3)
    elif case == "exact grid (all co-circular)":
        xs, ys = np.meshgrid(np.arange(90, dtype=np.float32) * 4, np.arange(60, dtype=np.float32) * 4)
        pos = np.stack([xs.ravel(), ys.ravel()], 1)
    elif case == "clustered + sparse":
        a = rng.normal([100, 100], 15, (3000, 2))
        b = rng.normal([500, 300], 40, (2500, 2))
        c = rng.random((300, 2)) * [640, 480]
        pos = np.concatenate([a, b, c]).astype(np.float32)
    else:
        g = synth.make_points(640, 480, 6, 9)
        rows = np.stack([np.arange(400, dtype=np.float32) * 1.5, np.full(400, 240.0, np.float32)], 1)
        pos = np.concatenate([g, g[::17], rows]).astype(np.float32)
    seq = _triangulate_in_subprocess(pos, 1, 1)
    par1 = _triangulate_in_subprocess(pos, 0, 1)
    par4 = _triangulate_in_subprocess(pos, 0, 4)
    many = _triangulate_in_subprocess(pos, 24, 3)
    assert np.array_equal(par1[0], par4[0]) and np.array_equal(par1[1], par4[1]), "output depends on the thread count"
    if case != "exact grid (all co-circular)":  # (a co-circular set has many Delaunay triangulations: only validity is comparable)
        assert _sets(*seq) == _sets(*par1) == _sets(*many)
    for tris, edges in (seq, par1, many):
        n_used = len(np.unique(tris))
        hull = 3 * n_used - 3 - len(edges)
        assert len(tris) == 2 * n_used - 2 - hull and hull >= 3
        p = pos.astype(np.float64)
        a, b, c = p[tris[:, 0]], p[tris[:, 1]], p[tris[:, 2]]
        assert np.all((b[:, 0] - a[:, 0]) * (c[:, 1] - a[:, 1]) - (b[:, 1] - a[:, 1]) * (c[:, 0] - a[:, 0]) > 0)
        assert len(_sets(tris, edges)[1]) == len(edges)


def _locally_delaunay_exact(pos, tris):
    """Every interior edge of the triangulation passes the in-circle test with the vertex across it -- in exact integer arithmetic
    (the coordinates are small integers)."""
    p = np.asarray(pos, dtype=np.int64)
    t = np.asarray(tris, dtype=np.int64)
    owner = {}
    for ti, (a, b, c) in enumerate(t.tolist()):
        for u, v, w in ((a, b, c), (b, c, a), (c, a, b)):
            owner[(u, v)] = (ti, w)
    bad = 0
    for (u, v), (ti, w) in owner.items():
        other = owner.get((v, u))
        if other is None:
            continue
        d = other[1]
        ax, ay = (p[u] - p[d]).tolist()
        bx, by = (p[v] - p[d]).tolist()
        cx, cy = (p[w] - p[d]).tolist()
        det = (ax * ax + ay * ay) * (bx * cy - cx * by) + (bx * bx + by * by) * (cx * ay - ax * cy) + (cx * cx + cy * cy) * (ax * by - bx * ay)
        bad += det > 0  # d strictly inside the circle through u, v, w (counter-clockwise)
    return bad == 0


@pytest.mark.parametrize("seed", [1, 2, 3, 4])
def test_merged_strips_on_degenerate_inputs_are_delaunay(built, seed):
    """Round 5: strips of their own points merged along their seams (lower tangent, zipper over the hulls' ghost triangles, Lawson
    flips).  Integer pixel positions -- collinear runs on the hulls and along the seams, co-circular quadruples everywhere -- have
    many Delaunay triangulations: whatever the merge returns must BE one (exact in-circle test across every interior edge, every
    triangle counter-clockwise, Euler), with the counts of the sequential build; for 2 ... 32 strips."""
    from flame_amd.regularizer import delaunay

    rng = np.random.default_rng(seed)
    n = (5000, 7000, 4500, 9000)[seed - 1]
    w, h = ((640, 480), (400, 300), (1280, 96), (320, 240))[seed - 1]   # (seed 3: strips much taller than wide would be; seed 4: dense)
    pos = np.unique(rng.integers(0, [w, h], (n, 2)), axis=0).astype(np.float32)
    if seed == 2:  # plus full rows and columns: long collinear runs across every seam
        rows = np.stack([np.arange(w, dtype=np.float32), np.full(w, 150.0, np.float32)], 1)
        cols = np.stack([np.full(h, 200.0, np.float32), np.arange(h, dtype=np.float32)], 1)
        pos = np.unique(np.concatenate([pos, rows, cols]), axis=0).astype(np.float32)
    if len(pos) < 4096:
        pos = np.unique(np.concatenate([pos, rng.integers(0, [w, h], (6000, 2)).astype(np.float32)]), axis=0)
    old = {k: os.environ.get(k) for k in ("FLAME_DELAUNAY_STRIPS", "FLAME_DELAUNAY_MERGE")}
    try:
        os.environ["FLAME_DELAUNAY_STRIPS"] = "1"
        t_seq, e_seq = delaunay(pos)
        for strips in (2, 7, 16, 32):
            os.environ["FLAME_DELAUNAY_STRIPS"], os.environ["FLAME_DELAUNAY_MERGE"] = str(strips), "1"
            t, e = delaunay(pos)
            assert len(t) == len(t_seq) and len(e) == len(e_seq), (strips, len(t), len(t_seq))
            p = pos.astype(np.float64)
            a, b, c = p[t[:, 0]], p[t[:, 1]], p[t[:, 2]]
            assert np.all((b[:, 0] - a[:, 0]) * (c[:, 1] - a[:, 1]) - (b[:, 1] - a[:, 1]) * (c[:, 0] - a[:, 0]) > 0)
            assert len({(min(u, v), max(u, v)) for u, v in e.tolist()}) == len(e)
            assert _locally_delaunay_exact(pos, t), strips
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


@pytest.mark.parametrize("name", ["jittered_640x480", "uniform_6000", "clustered_5400"])
def test_strip_paths_equal_reference_triangle_run_on_large_sets(built, name):
    """The strip paths (>= 4096 points: merged strips, the default since round 5; certified strips) against the REFERENCE's Triangle run
    here (oracle/make_golden.py: make_large_delaunay_fixture, oracle/_ref): the sorted triangle set and the sorted edge set, by their
    SHA-256 -- the point sets are regenerated from their seeds."""
    import json

    from flame_amd.regularizer import delaunay
    from oracle.make_golden import canonical_digests, large_delaunay_sets

    want = json.load(open(os.path.join(GOLDEN, "delaunay_ref_triangle_large.json")))[name]
    pts = large_delaunay_sets()[name]
    assert len(pts) == want["points"]
    old = {k: os.environ.get(k) for k in ("FLAME_DELAUNAY_STRIPS", "FLAME_DELAUNAY_MERGE")}
    try:
        for strips, merge in ((None, "1"), ("7", "1"), ("32", "1"), (None, "0"), ("1", "1")):
            os.environ["FLAME_DELAUNAY_MERGE"] = merge
            if strips is None:
                os.environ.pop("FLAME_DELAUNAY_STRIPS", None)
            else:
                os.environ["FLAME_DELAUNAY_STRIPS"] = strips
            t, e = delaunay(pts)
            assert (len(t), len(e)) == (want["triangles"], want["edges"]), (strips, merge)
            ht, he = canonical_digests(t, e)
            assert ht == want["triangles_sha256"] and he == want["edges_sha256"], (strips, merge)
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def test_the_library_own_delaunay_check_accepts_the_merged_strips(built):
    """FLAME_DELAUNAY_VERIFY=1 (advisor, round 5): the library itself runs the exact in-circle predicate across every interior edge of a
    merged triangulation and falls back to the certified strips if one fails.  On the jittered 640x480 set and on a co-circular
    integer grid the merged result passes -- no fall-back line -- and is the same set of triangles as without the switch."""
    import subprocess
    import sys

    code = r"""
import os, sys
sys.path.insert(0, {root!r})
import numpy as np
from flame_amd import synth
from flame_amd.regularizer import delaunay
pos = np.ascontiguousarray(synth.make_graph("640x480", seed=3)["pos"], np.float32)
grid = np.stack(np.meshgrid(np.arange(90, dtype=np.float32), np.arange(60, dtype=np.float32)), -1).reshape(-1, 2)
for name, p in (("jittered", pos), ("grid", grid)):
    os.environ.pop("FLAME_DELAUNAY_VERIFY", None)
    t0, e0 = delaunay(p)
    os.environ["FLAME_DELAUNAY_VERIFY"] = "1"
    t1, e1 = delaunay(p)
    same = sorted(map(tuple, np.sort(t0, 1).tolist())) == sorted(map(tuple, np.sort(t1, 1).tolist()))
    print(name, len(p), len(t1), len(e1), "same" if same and len(e0) == len(e1) else "DIFFERENT")
"""
    from tests.conftest import ROOT

    env = dict(os.environ, FLAME_DELAUNAY_TRACE="1")
    r = subprocess.run([sys.executable, "-c", code.format(root=ROOT)], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    assert r.stdout.count("same") == 2 and "DIFFERENT" not in r.stdout, r.stdout
    assert "falling back" not in r.stderr and "gave up" not in r.stderr, r.stderr[-2000:]
