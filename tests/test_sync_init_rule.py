"""The neighbour-mean initialisation of new vertices (init_with_prediction's fallback, flame.cc:2133-2158) and its one
stated deviation from the reference.

The reference initialises new vertices one after the other, in the iteration order of an unordered_set (feats_to_update)
and of each vertex's hash-set adjacency: a new vertex that is initialised later sees the value an adjacent new vertex was
given just before (Gauss-Seidel), and which of two adjacent new vertices comes first is unspecified.  The library (and its
checker oracle/sync_oracle.py) fix an order-free rule instead (Jacobi): every new vertex without a prediction takes the mean
over its neighbours of the values standing BEFORE any of them was initialised (such a neighbour stands at its data term,
flame.cc:2046-2048).  The two rules agree exactly unless two new vertices WITHOUT a prediction are adjacent; this test pins
both statements against an explicit sequential restatement of the reference's loop (advisor finding r02,
nltgv2_layout.hip:494), so the deviation is visible and bounded instead of implicit."""
import numpy as np
import pytest

from flame_amd import synth
from oracle import sync_oracle

F = np.float32
GS = F(1.3)


def sequential_reference(prev_x, nbrs, weight, data, order):
    """flame.cc:2133-2158 on explicit orders: for f in `order`: x[f] = (sum over neighbours with weight > 0 of x * scale)
    / count / scale, neighbours in the given adjacency order, using the CURRENT x of every neighbour."""
    x = dict(prev_x)
    for f in order:
        s, cnt = F(0), 0
        for nb in nbrs[f]:
            if weight[nb] > 0:
                s = F(s + F(x[nb] * GS))
                cnt += 1
        x[f] = F(F(s / F(cnt)) / GS) if cnt else F(data[f])
    return x


def build(adjacent: bool):
    """Six old vertices on a 3 x 2 grid (features 0..5) and two new ones (features 10, 11) without a prediction; with
    `adjacent` the two new vertices share an edge."""
    pos_old = np.array([[0, 0], [6, 0], [12, 0], [0, 6], [6, 6], [12, 6]], F)
    edges_old = np.array([[0, 1], [1, 2], [3, 4], [4, 5], [0, 3], [1, 4], [2, 5], [0, 4], [1, 5]], np.int32)
    g0 = synth.assemble_graph(pos_old, np.array([0.8, 0.9, 1.0, 1.1, 1.2, 1.3], F), edges_old)
    g0["x"] = g0["x_bar"] = g0["x_prev"] = np.array([0.81, 0.93, 1.02, 1.08, 1.22, 1.27], F)
    ref = sync_oracle.RefGraph.from_flat(g0, np.arange(6, dtype=np.int32))
    feat = np.array([0, 1, 2, 3, 4, 5, 10, 11], np.int32)
    pos = np.concatenate([pos_old, np.array([[3, 12], [9, 12]], F)])
    data = np.array([0.8, 0.9, 1.0, 1.1, 1.2, 1.3, 2.0, 0.5], F)
    weight = np.ones(8, F)
    new_edges = [[3, 6], [4, 6], [4, 7], [5, 7]] + ([[6, 7]] if adjacent else [])
    edges = np.concatenate([edges_old, np.array(new_edges, np.int32)])
    init_x = np.array([np.nan] * 8, F)  # (only read for the new vertices)
    return ref, feat, pos, data, weight, edges, init_x


@pytest.mark.parametrize("adjacent", [False, True])
def test_jacobi_rule_against_the_sequential_reference(adjacent):
    ref, feat, pos, data, weight, edges, init_x = build(adjacent)
    out = sync_oracle.sync(ref, feat, pos, data, weight, edges, init_x=init_x, init_graph_scale=float(GS))
    got = {f: F(out.v[f]["x"]) for f in (10, 11)}
    # the state the sequential loop starts from: old vertices keep x, new ones stand at their data term
    prev_x = {f: F(out.v[f]["x"]) for f in range(6)}
    prev_x.update({10: F(2.0), 11: F(0.5)})
    w = {int(f): float(weight[i]) for i, f in enumerate(feat)}
    d = {int(f): F(data[i]) for i, f in enumerate(feat)}
    idx = {int(f): i for i, f in enumerate(feat)}
    nbrs = {10: [], 11: []}
    for a, b in edges:  # ascending edge id, as the checker walks them
        fa, fb = int(feat[a]), int(feat[b])
        if fa in nbrs:
            nbrs[fa].append(fb)
        if fb in nbrs:
            nbrs[fb].append(fa)
    assert idx[10] == 6 and idx[11] == 7
    seq_a = sequential_reference(prev_x, nbrs, w, d, order=[10, 11])
    seq_b = sequential_reference(prev_x, nbrs, w, d, order=[11, 10])
    if not adjacent:
        # no two unpredicted new vertices touch: the library's rule IS the reference's loop, whatever its order -- bit for bit
        assert got[10].tobytes() == seq_a[10].tobytes() == seq_b[10].tobytes()
        assert got[11].tobytes() == seq_a[11].tobytes() == seq_b[11].tobytes()
    else:
        # adjacent: the reference's result depends on its (unspecified) order -- the FIRST vertex of either order gets exactly
        # the library's value, the second one differs by what the first one moved by, divided by its neighbour count
        assert got[10].tobytes() == seq_a[10].tobytes() and got[11].tobytes() == seq_b[11].tobytes()
        assert seq_a[11] != got[11] and seq_b[10] != got[10]
        assert abs(float(seq_a[11]) - float(got[11])) == pytest.approx(abs(float(seq_a[10]) - 2.0) / 3, rel=1e-5)
        assert abs(float(seq_b[10]) - float(got[10])) == pytest.approx(abs(float(seq_b[11]) - 0.5) / 3, rel=1e-5)
        assert seq_a[11] != seq_b[11]  # ... and the reference's two orders disagree with each other


@pytest.mark.gpu
@pytest.mark.parametrize("adjacent", [False, True])
def test_device_follows_the_jacobi_rule(built, adjacent):
    import torch  # noqa: F401

    import flame_amd

    ref, feat, pos, data, weight, edges, init_x = build(adjacent)
    g0 = sync_oracle.flatten(ref, np.arange(6, dtype=np.int32))
    with flame_amd.Regularizer(0) as reg:
        reg.upload_graph(g0)
        reg.set_feature_ids(np.arange(6, dtype=np.int32))
        reg.sync_graph(feat, pos, data, weight, edges, init_x=init_x, init_graph_scale=float(GS))
        x = reg.download_state(("x",))["x"]
    out = sync_oracle.sync(ref, feat, pos, data, weight, edges, init_x=init_x, init_graph_scale=float(GS))
    want = np.array([out.v[int(f)]["x"] for f in feat], F)
    assert np.array_equal(x, want)
