"""Per-frame warm-start graph synchronisation: flame_nltgv2_sync_graph against the
restatement of Flame::syncGraph's graph edits (oracle/sync_oracle.py) over a sequence of frames with
vertex churn, re-projection, re-triangulation and the sticky-obstacle reset -- every state array and the
resulting edge list must match exactly, frame after frame."""
import numpy as np
import pytest

from flame_amd import synth
from oracle import capi as oracle
from oracle import sync_oracle
from tests.helpers import OUT_KEYS, assert_state_equal

pytestmark = pytest.mark.gpu


def next_frame(rng, feat_id, pos, data, next_id, width, height):
    """Drops ~8 % of the features, jitters the survivors (projectGraph), adds ~8 % new ones."""
    keep = rng.random(len(feat_id)) > 0.08
    feat_id, pos, data = feat_id[keep], pos[keep].copy(), data[keep].copy()
    pos += rng.normal(0, 0.4, pos.shape).astype(np.float32)
    data = (data * np.float32(1.0) + rng.normal(0, 0.01, data.shape)).astype(np.float32)
    n_new = int(0.08 * len(keep))
    new_pos = np.stack([rng.random(n_new) * (width - 8) + 4, rng.random(n_new) * (height - 8) + 4], 1).astype(np.float32)
    new_data = (0.5 + rng.random(n_new)).astype(np.float32)
    new_id = np.arange(next_id, next_id + n_new)
    order = rng.permutation(len(feat_id) + n_new)  # the caller's vertex order changes freely
    feat_id = np.concatenate([feat_id, new_id])[order]
    pos = np.concatenate([pos, new_pos])[order]
    data = np.concatenate([data, new_data])[order]
    return feat_id.astype(np.int32), np.ascontiguousarray(pos), np.ascontiguousarray(data), next_id + n_new


SYNC_MODES = ["host", "device"]  # index maps + layout tables on the host (rounds 1-3) / by kernels over the resident topology (round 4)


def _mode(reg, mode):
    """-> keyword arguments of sync_graph for the mode; the device path is asked for BY NAME, so a silent fall-back fails the test."""
    import flame_amd

    reg.set_option(flame_amd.regularizer.OPT_SYNC_PATH, 2 if mode == "device" else 1)
    return {"edges_unique": mode == "device"}


@pytest.mark.parametrize("mode", SYNC_MODES)
def test_frame_sequence_matches_syncgraph_restatement(built, mode):
    import torch  # noqa: F401

    import flame_amd

    rng = np.random.default_rng(7)
    g0 = synth.make_graph("320x240", seed=12)
    feat_id = np.arange(g0["V"], dtype=np.int32) * 3 + 5  # arbitrary stable ids
    pos, data = g0["pos"].copy(), g0["data_term"].copy()
    next_id = int(feat_id.max()) + 1
    params = flame_amd.Params()

    ref = sync_oracle.RefGraph.from_flat(g0, feat_id)
    with flame_amd.Regularizer(0) as reg:
        reg.upload_graph(g0)
        reg.set_feature_ids(feat_id)
        kw = _mode(reg, mode)
        for frame in range(5):
            # solve a little on both sides
            flat = sync_oracle.flatten(ref, feat_id)
            src, dst, fid = reg.topology()
            assert np.array_equal(fid, feat_id)
            assert np.array_equal(src, flat["src"]) and np.array_equal(dst, flat["dst"]), f"frame {frame}: edge list"
            n = 25 + 3 * frame
            reg.run(params, n)
            assert oracle.run(flat, n) == 0
            out = reg.download_state()
            assert_state_equal(out, flat, keys=OUT_KEYS + ("x_prev", "w1_prev", "w2_prev"), what=f"frame {frame}")
            sync_oracle.absorb(ref, flat, feat_id)
            # next frame: churn, re-projection, new triangulation
            feat_id, pos, data, next_id = next_frame(rng, feat_id, pos, data, next_id, 320, 240)
            weight = (0.5 + rng.random(len(feat_id))).astype(np.float32)
            edges = synth.delaunay_edges_scipy(pos)
            if frame % 2:
                edges = edges[:, ::-1].copy()  # the triangulator's orientation is arbitrary; survivors keep theirs
            init_x = (data * np.float32(1.02)).astype(np.float32)
            sticky = frame >= 2
            gs = 0.0
            if frame % 2:  # no valid prediction for a third of the vertices: neighbours' mean (flame.cc:2133-2158)
                init_x[rng.random(len(feat_id)) < 0.33] = np.nan
                weight[rng.random(len(feat_id)) < 0.1] = 0.0  # ... over the neighbours with a data weight
                gs = 1.7
            reg.sync_graph(feat_id, pos, data, weight, edges, init_x=init_x, check_sticky_obstacles=sticky,
                           sticky_threshold=0.02, init_graph_scale=gs, **kw)
            assert reg.info()["last_sync_path"] == (2 if mode == "device" else 1)
            assert reg.layout_selftest() == 0, f"frame {frame}: device tables differ from the host builders'"
            sync_oracle.sync(ref, feat_id, pos, data, weight, edges, init_x=init_x, check_sticky=sticky, thr=0.02,
                             init_graph_scale=gs)
        flat = sync_oracle.flatten(ref, feat_id)
        assert_state_equal(reg.download_state(), flat, keys=OUT_KEYS + ("x_prev",), what="after last sync")


@pytest.mark.parametrize("mode", SYNC_MODES)
def test_feature_ids_anywhere_in_int32(built, mode):
    """The reference's feature ids grow by one per detection for the whole session (millions within minutes); the sync must not
    depend on their magnitude.  Ids scattered over [0, 2^31): a few small ones, most beyond any direct table, new ones up to
    INT32_MAX -- the device path keeps them in two stamped hash tables (nltgv2_topo.hip: feat_insert / feat_lookup) and stays the
    path taken; every frame equal to the restatement."""
    import torch  # noqa: F401

    import flame_amd

    rng = np.random.default_rng(77)
    g0 = synth.make_graph("320x240", seed=13)
    V = g0["V"]
    feat_id = rng.choice(np.arange(5_000_000, 2_147_000_000, 997, dtype=np.int64), size=V, replace=False).astype(np.int32)
    feat_id[:16] = np.arange(16, dtype=np.int32) * 7  # (and some small ones)
    pos, data = g0["pos"].copy(), g0["data_term"].copy()
    params = flame_amd.Params()
    ref = sync_oracle.RefGraph.from_flat(g0, feat_id)
    with flame_amd.Regularizer(0) as reg:
        reg.upload_graph(g0)
        reg.set_feature_ids(feat_id)
        kw = _mode(reg, mode)
        for frame in range(4):
            flat = sync_oracle.flatten(ref, feat_id)
            reg.run(params, 20)
            assert oracle.run(flat, 20) == 0
            assert_state_equal(reg.download_state(), flat, keys=OUT_KEYS, what=f"frame {frame}")
            sync_oracle.absorb(ref, flat, feat_id)
            keep = rng.random(len(feat_id)) > 0.1
            n_new = int((~keep).sum())
            new_id = np.setdiff1d(rng.choice(np.arange(1000, 2_147_483_647, 1013, dtype=np.int64), size=2 * n_new, replace=False), feat_id)[:n_new]
            if frame == 2:
                new_id[0] = 2_147_483_647  # INT32_MAX itself
            order = rng.permutation(len(feat_id))
            feat_id = np.concatenate([feat_id[keep], new_id]).astype(np.int32)[order]
            pos = np.concatenate([pos[keep] + rng.normal(0, 0.3, (int(keep.sum()), 2)).astype(np.float32),
                                  np.stack([rng.random(n_new) * 312 + 4, rng.random(n_new) * 232 + 4], 1).astype(np.float32)])[order]
            data = np.concatenate([data[keep], (0.5 + rng.random(n_new)).astype(np.float32)])[order]
            pos, data = np.ascontiguousarray(pos, np.float32), np.ascontiguousarray(data, np.float32)
            weight = np.ones(len(feat_id), np.float32)
            edges = synth.delaunay_edges_scipy(pos)
            reg.sync_graph(feat_id, pos, data, weight, edges, **kw)
            assert reg.info()["last_sync_path"] == (2 if mode == "device" else 1)
            sync_oracle.sync(ref, feat_id, pos, data, weight, edges)
        flat = sync_oracle.flatten(ref, feat_id)
        assert_state_equal(reg.download_state(), flat, keys=OUT_KEYS, what="after the last sync")
        dup = feat_id.copy()
        dup[3] = dup[9]
        with pytest.raises(flame_amd.NLTGV2Error):  # a duplicate id is still an invalid argument, whatever its magnitude
            reg.sync_graph(dup, pos, data, weight, edges, **kw)


def test_sync_rejects_bad_input(built):
    import torch  # noqa: F401

    import flame_amd

    g = synth.make_graph("320x240", seed=1)
    with flame_amd.Regularizer(0) as reg:
        reg.upload_graph(g)
        fid = np.arange(g["V"], dtype=np.int32)
        fid[3] = fid[2]  # duplicate feature id
        with pytest.raises(flame_amd.NLTGV2Error):
            reg.sync_graph(fid, g["pos"], g["data_term"], g["data_weight"], np.stack([g["src"], g["dst"]], 1))
        with pytest.raises(flame_amd.NLTGV2Error):
            reg.sync_graph(np.arange(g["V"], dtype=np.int32), g["pos"], g["data_term"], g["data_weight"],
                           np.array([[0, g["V"]]], np.int32))


@pytest.mark.parametrize("mode", SYNC_MODES)
@pytest.mark.parametrize("config", ["320x240", "640x480", "1280x720", "1920x1080"])
def test_device_expanded_layout_matches_host_builders(built, config, mode):
    """Per-slot (B) and per-lane (E) layout arrays are expanded on the device (nltgv2_layout.hip): word for word what the
    host builders of nltgv2_pack.hpp make of the same topology -- after an upload and after a frame sync."""
    import torch  # noqa: F401

    import flame_amd

    g = synth.make_graph(config, seed=3)
    with flame_amd.Regularizer(0) as reg:
        reg.upload_graph(g)
        assert reg.layout_selftest() == 0
        rng = np.random.default_rng(5)
        w, h = (int(s) for s in config.split("x"))
        fid, pos, data, _ = next_frame(rng, np.arange(g["V"], dtype=np.int32), g["pos"].copy(), g["data_term"].copy(), g["V"], w, h)
        reg.sync_graph(fid, pos, data, np.ones(len(fid), np.float32), synth.delaunay_edges_scipy(pos), **_mode(reg, mode))
        assert reg.info()["last_sync_path"] == (2 if mode == "device" else 1)
        assert reg.layout_selftest() == 0
        reg.run(flame_amd.Params(), 16)
        assert reg.layout_selftest() == 0
        reg.set_option(flame_amd.regularizer.OPT_PERSISTENT, 3)  # the vertex-per-lane rows: built on demand
        reg.run(flame_amd.Params(), 16)
        assert reg.info()["last_run_path"] == 5 and reg.layout_selftest() == 0


@pytest.mark.parametrize("mode", SYNC_MODES)
def test_sync_behind_an_unchecked_chain_and_with_growing_graphs(built, mode):
    """sync_graph settles a chain of asynchronous runs first; consecutive syncs without a run in between; the graph grows
    past every buffer's capacity and shrinks again; a rejected sync leaves the previous graph intact."""
    import torch  # noqa: F401

    import flame_amd

    rng = np.random.default_rng(11)
    params = flame_amd.Params()
    g0 = synth.make_graph("320x240", seed=5)
    feat_id = np.arange(g0["V"], dtype=np.int32)
    ref = sync_oracle.RefGraph.from_flat(g0, feat_id)

    def both_run(reg, n, asynchronous=False):
        flat = sync_oracle.flatten(ref, feat_id)
        assert oracle.run(flat, n) == 0
        sync_oracle.absorb(ref, flat, feat_id)
        (reg.run_async if asynchronous else reg.run)(params, n)

    with flame_amd.Regularizer(0) as reg:
        reg.upload_graph(g0)
        reg.set_feature_ids(feat_id)
        pos, data, next_id = g0["pos"].copy(), g0["data_term"].copy(), g0["V"]
        kw = _mode(reg, mode)
        both_run(reg, 12, asynchronous=True)
        both_run(reg, 9, asynchronous=True)   # a chain: the second run sits behind an unchecked persistent run
        for frame, (w, h) in enumerate([(320, 240), (640, 480), (640, 480), (320, 240)]):
            if frame in (1, 3):  # a much larger / smaller frame: every vertex is new, every buffer is re-sized
                g = synth.make_graph(f"{w}x{h}", seed=20 + frame)
                feat_id = np.arange(next_id, next_id + g["V"], dtype=np.int32)
                pos, data, next_id = g["pos"].copy(), g["data_term"].copy(), next_id + g["V"]
            else:
                feat_id, pos, data, next_id = next_frame(rng, feat_id, pos, data, next_id, w, h)
            weight = np.ones(len(feat_id), np.float32)
            edges = synth.delaunay_edges_scipy(pos)
            reg.sync_graph(feat_id, pos, data, weight, edges, **kw)
            sync_oracle.sync(ref, feat_id, pos, data, weight, edges)
            if frame == 2:   # twice in a row, nothing run in between (the state is in the canonical arrays)
                feat_id, pos, data, next_id = next_frame(rng, feat_id, pos, data, next_id, w, h)
                weight = np.ones(len(feat_id), np.float32)
                edges = synth.delaunay_edges_scipy(pos)
                reg.sync_graph(feat_id, pos, data, weight, edges, **kw)
                sync_oracle.sync(ref, feat_id, pos, data, weight, edges)
            bad = feat_id.copy()
            bad[1] = bad[0]
            with pytest.raises(flame_amd.NLTGV2Error):
                reg.sync_graph(bad, pos, data, weight, edges, **kw)  # refused: the graph of this frame stays
            both_run(reg, 20)
            assert_state_equal(reg.download_state(), sync_oracle.flatten(ref, feat_id),
                               keys=OUT_KEYS + ("x_prev", "w1_prev", "w2_prev"), what=f"frame {frame}")
            assert reg.layout_selftest() == 0


def test_sync_to_empty_edgeless_and_single_vertex_frames(built):
    """Degenerate frames: no vertices at all, vertices without edges, one vertex -- and back to a full frame (all new)."""
    import torch  # noqa: F401

    import flame_amd

    p = flame_amd.Params()
    g = synth.make_graph("320x240", seed=2)
    e0 = np.zeros((0, 2), np.int32)
    fid = np.arange(g["V"], dtype=np.int32)
    with flame_amd.Regularizer(0) as r:
        r.upload_graph(g)
        r.run(p, 10)
        r.sync_graph(np.zeros(0, np.int32), np.zeros((0, 2), np.float32), np.zeros(0, np.float32), np.zeros(0, np.float32), e0)
        assert (r.info()["V"], r.info()["E"]) == (0, 0)
        r.run(p, 5)
        assert r.download_state()["x"].shape == (0,)
        r.sync_graph(fid, g["pos"], g["data_term"], g["data_weight"], np.stack([g["src"], g["dst"]], 1))
        r.run(p, 10)
        ref = {k: (v.copy() if hasattr(v, "copy") else v) for k, v in g.items()}
        assert oracle.run(ref, 10) == 0
        assert_state_equal(r.download_state(), ref, what="empty -> full frame")
        r.sync_graph(fid[:50], g["pos"][:50], g["data_term"][:50], g["data_weight"][:50], e0)
        r.run(p, 7)
        assert (r.info()["V"], r.info()["E"]) == (50, 0) and r.layout_selftest() == 0
        r.sync_graph(fid[:1], g["pos"][:1], g["data_term"][:1], g["data_weight"][:1], e0)
        r.run(p, 3)
        assert r.download_state()["x"].shape == (1,)


@pytest.mark.parametrize("mode", SYNC_MODES)
@pytest.mark.parametrize("seed", [1, 2])
def test_randomized_frame_sequences(built, seed, mode):
    """Thirty frames of random churn (0-60 % of the vertices replaced), random triangulator orientation, zero data weights,
    missing predictions (neighbour-mean init), sticky obstacles, random run lengths, synchronous and chained asynchronous
    runs: every state array and the edge list equal the syncGraph restatement's after every frame."""
    import torch  # noqa: F401

    import flame_amd

    rng = np.random.default_rng(100 + seed)
    g0 = synth.make_graph("320x240", seed=30 + seed)
    feat_id = (np.arange(g0["V"], dtype=np.int32) * 7 + 3)
    pos, data = g0["pos"].copy(), g0["data_term"].copy()
    next_id = int(feat_id.max()) + 1
    params = flame_amd.Params()
    ref = sync_oracle.RefGraph.from_flat(g0, feat_id)
    with flame_amd.Regularizer(0) as reg:
        reg.upload_graph(g0)
        reg.set_feature_ids(feat_id)
        kw = _mode(reg, mode)
        for frame in range(30):
            flat = sync_oracle.flatten(ref, feat_id)
            n_total = 0
            for _ in range(int(rng.integers(1, 4))):
                n = int(rng.integers(1, 40))
                (reg.run_async if rng.random() < 0.5 else reg.run)(params, n)
                n_total += n
            assert oracle.run(flat, n_total) == 0
            sync_oracle.absorb(ref, flat, feat_id)
            # next frame
            keep = rng.random(len(feat_id)) > rng.random() * 0.6
            keep[: 3] = True
            n_new = int(rng.integers(0, 200))
            new_pos = np.stack([rng.random(n_new) * 312 + 4, rng.random(n_new) * 232 + 4], 1).astype(np.float32)
            order = rng.permutation(int(keep.sum()) + n_new)
            feat_id = np.concatenate([feat_id[keep], np.arange(next_id, next_id + n_new)])[order].astype(np.int32)
            next_id += n_new
            pos = np.concatenate([(pos[keep] + rng.normal(0, 0.4, (int(keep.sum()), 2))).astype(np.float32), new_pos])[order]
            pos = np.ascontiguousarray(np.clip(pos, 1.0, [318.0, 238.0]).astype(np.float32))
            data = np.concatenate([data[keep], (0.5 + rng.random(n_new)).astype(np.float32)])[order].astype(np.float32)
            weight = (0.5 + rng.random(len(feat_id))).astype(np.float32)
            weight[rng.random(len(feat_id)) < 0.1] = 0.0
            edges = synth.delaunay_edges_scipy(pos)
            if rng.random() < 0.5:
                edges = edges[:, ::-1].copy()
            init_x = (data * np.float32(1.0 + 0.05 * rng.random())).astype(np.float32)
            gs = 0.0
            if rng.random() < 0.5:
                init_x[rng.random(len(feat_id)) < 0.3] = np.nan
                gs = float(0.5 + rng.random() * 2)
            sticky = bool(rng.random() < 0.5)
            reg.sync_graph(feat_id, pos, data, weight, edges, init_x=init_x, check_sticky_obstacles=sticky, sticky_threshold=0.03,
                           init_graph_scale=gs, **kw)
            sync_oracle.sync(ref, feat_id, pos, data, weight, edges, init_x=init_x, check_sticky=sticky, thr=0.03, init_graph_scale=gs)
            flat = sync_oracle.flatten(ref, feat_id)
            src, dst, fid = reg.topology()
            assert np.array_equal(src, flat["src"]) and np.array_equal(dst, flat["dst"]) and np.array_equal(fid, feat_id), frame
            assert_state_equal(reg.download_state(), flat, keys=OUT_KEYS + ("x_prev", "w1_prev", "w2_prev"), what=f"seed {seed} frame {frame}")


def test_edges_unique_flag_skips_nothing_but_the_search(built):
    """flame_nltgv2_sync_input.edges_unique: with a duplicate-free edge list (a triangulator's) the flag only skips the search for
    repeated pairs -- the resulting edge list and state are those of the default path."""
    import torch  # noqa: F401

    import flame_amd

    rng = np.random.default_rng(3)
    g0 = synth.make_graph("320x240", seed=5)
    feat_id = np.arange(g0["V"], dtype=np.int32)
    f1, p1, d1, _ = next_frame(rng, feat_id, g0["pos"].copy(), g0["data_term"].copy(), g0["V"], 320, 240)
    edges = synth.delaunay_edges_native(p1)
    ones = np.ones(len(f1), np.float32)
    outs = []
    for flag in (False, True):
        with flame_amd.Regularizer(0) as reg:
            reg.upload_graph(g0)
            reg.set_feature_ids(feat_id)
            reg.run(flame_amd.Params(), 30)
            reg.sync_graph(f1, p1, d1, ones, edges, edges_unique=flag)
            reg.run(flame_amd.Params(), 30)
            outs.append((reg.topology(), reg.download_state()))
    (ta, sa), (tb, sb) = outs
    assert all(np.array_equal(x, y) for x, y in zip(ta, tb))
    assert_state_equal(sa, sb, keys=OUT_KEYS, what="edges_unique")


@pytest.mark.parametrize("config", ["320x240", "640x480"])
def test_prepared_sync_beside_a_running_solver(built, config):
    """flame_nltgv2_sync_prepare / _commit: the new frame's graph is built on a side stream while the solver keeps iterating on the
    old one (run_async between the two halves); commit moves the state as it stands THEN.  Eight frames, every state array and the
    edge list equal the syncGraph restatement's (which solves first, then syncs) -- plus the cases in which a prepared sync goes the
    host way or is cancelled."""
    import torch  # noqa: F401

    import flame_amd

    w, h = (int(s) for s in config.split("x"))
    rng = np.random.default_rng(21)
    g0 = synth.make_graph(config, seed=8)
    feat_id = np.arange(g0["V"], dtype=np.int32) * 2 + 1
    pos, data = g0["pos"].copy(), g0["data_term"].copy()
    next_id = int(feat_id.max()) + 1
    params = flame_amd.Params()
    ref = sync_oracle.RefGraph.from_flat(g0, feat_id)

    def ref_run(ids, n):
        flat = sync_oracle.flatten(ref, ids)
        assert oracle.run(flat, n) == 0
        sync_oracle.absorb(ref, flat, ids)

    with flame_amd.Regularizer(0) as reg:
        reg.upload_graph(g0)
        reg.set_feature_ids(feat_id)
        reg.run(params, 20)
        ref_run(feat_id, 20)
        for frame in range(8):
            old_ids = feat_id
            feat_id, pos, data, next_id = next_frame(rng, feat_id, pos, data, next_id, w, h)
            weight = (0.5 + rng.random(len(feat_id))).astype(np.float32)
            edges = synth.delaunay_edges_scipy(pos)
            init_x = (data * np.float32(1.01)).astype(np.float32)
            gs = 0.0
            if frame % 3 == 1:
                init_x[rng.random(len(feat_id)) < 0.3] = np.nan
                gs = 1.3
            unique = frame != 5           # frame 5: no vouched-for edge list -> the host way at commit
            fid_arg, pos_arg = feat_id.copy(), pos.copy()
            reg.sync_prepare(fid_arg, pos_arg, data, weight, edges, init_x=init_x, check_sticky_obstacles=frame % 2 == 0,
                             sticky_threshold=0.02, init_graph_scale=gs, edges_unique=unique)
            fid_arg[:] = -7               # the caller's arrays are free once prepare returns
            pos_arg[:] = np.nan
            n_between = 0
            for n in (17, 30, 8)[: 1 + frame % 3]:   # the solver keeps iterating on the OLD graph meanwhile
                reg.run_async(params, n)
                n_between += n
            reg.sync_commit()
            assert reg.info()["last_sync_path"] == (2 if unique else 1), frame
            ref_run(old_ids, n_between)
            sync_oracle.sync(ref, feat_id, pos, data, weight, edges, init_x=init_x, check_sticky=frame % 2 == 0, thr=0.02, init_graph_scale=gs)
            flat = sync_oracle.flatten(ref, feat_id)
            src, dst, fid = reg.topology()
            assert np.array_equal(src, flat["src"]) and np.array_equal(dst, flat["dst"]) and np.array_equal(fid, feat_id), frame
            assert_state_equal(reg.download_state(), flat, keys=OUT_KEYS + ("x_prev", "w1_prev", "w2_prev"), what=f"frame {frame} after commit")
            assert reg.layout_selftest() == 0
            reg.run(params, 25)
            ref_run(feat_id, 25)
            assert_state_equal(reg.download_state(), sync_oracle.flatten(ref, feat_id), keys=OUT_KEYS, what=f"frame {frame} after 25 steps")
        # a commit with nothing prepared, and a prepared sync cancelled by an upload
        with pytest.raises(flame_amd.NLTGV2Error):
            reg.sync_commit()
        # bad inputs on the HOST way (no vouched-for edge list) are reported by prepare itself, as sync_graph reports them: a
        # duplicate feature id, an edge out of range, a self-loop -- and the graph is left as it was
        fb, pb, db, _ = next_frame(rng, feat_id, pos, data, next_id, w, h)
        eb = synth.delaunay_edges_scipy(pb)
        for what in ("duplicate id", "edge out of range", "self-loop"):
            f_bad, e_bad = fb.copy(), eb.copy()
            if what == "duplicate id":
                f_bad[3] = f_bad[2]
            elif what == "edge out of range":
                e_bad[5, 1] = len(fb)
            else:
                e_bad[7, 1] = e_bad[7, 0]
            with pytest.raises(flame_amd.NLTGV2Error):
                reg.sync_prepare(f_bad, pb, db, np.ones(len(fb), np.float32), e_bad, edges_unique=False)
        src, dst, fid = reg.topology()
        assert np.array_equal(fid, feat_id)
        f2, p2, d2, _ = next_frame(rng, feat_id, pos, data, next_id, w, h)
        reg.sync_prepare(f2, p2, d2, np.ones(len(f2), np.float32), synth.delaunay_edges_scipy(p2), edges_unique=True)
        reg.upload_graph(g0)
        with pytest.raises(flame_amd.NLTGV2Error):
            reg.sync_commit()
        reg.run(params, 10)
        chk = synth.copy_graph(g0)
        assert oracle.run(chk, 10) == 0
        assert_state_equal(reg.download_state(), chk, keys=OUT_KEYS, what="after the cancelled sync")
        # a frame the builder declines (a hub of 70 edges): prepared on the device, committed the host way
        hub_pos = np.concatenate([g0["pos"], np.array([[w / 2 + 0.37, h / 2 + 0.21]], np.float32)])
        hub = len(hub_pos) - 1
        e_all = synth.delaunay_edges_scipy(hub_pos)
        have = {tuple(sorted(e)) for e in e_all.tolist()}
        extra = [(int(v), hub) for v in rng.permutation(g0["V"])[:200] if tuple(sorted((int(v), hub))) not in have][:70]
        e_hub = np.concatenate([e_all, np.array(extra, np.int32)])
        fid_h = np.arange(len(hub_pos), dtype=np.int32)
        data_h = np.concatenate([g0["data_term"], [1.0]]).astype(np.float32)
        reg.set_feature_ids(np.arange(g0["V"], dtype=np.int32))
        ref2 = sync_oracle.RefGraph.from_flat(chk, np.arange(g0["V"], dtype=np.int32))
        reg.sync_prepare(fid_h, hub_pos, data_h, np.ones(len(fid_h), np.float32), e_hub, edges_unique=True)
        reg.sync_commit()
        assert reg.info()["last_sync_path"] == 1 and reg.info()["max_degree"] > 64
        sync_oracle.sync(ref2, fid_h, hub_pos, data_h, np.ones(len(fid_h), np.float32), e_hub)
        assert_state_equal(reg.download_state(), sync_oracle.flatten(ref2, fid_h), keys=OUT_KEYS, what="declined build, host way")


@pytest.mark.parametrize("mode", SYNC_MODES + ["prepared"])
def test_new_vertices_start_at_the_resident_dense_map(built, mode):
    """flame_nltgv2_sync_input.init_from_map (init_with_prediction, flame.cc:2131): the prediction of a new vertex is read ON THE
    DEVICE from the dense map the last interpolate_mesh left there -- idepthmap(pos.y + 0.5f, pos.x + 0.5f) / graph_scale, NaN (the
    neighbours' mean) where no triangle covered the pixel -- and equals what the host gather of rounds 1-3 fed in as init_x.  The
    map comes from interpolate_mesh_begin / _end with the solver iterating in between."""
    import torch  # noqa: F401

    import flame_amd

    w, h = 320, 240
    rng = np.random.default_rng(33)
    g0 = synth.make_graph("320x240", seed=14)
    feat_id = np.arange(g0["V"], dtype=np.int32)
    pos, data = g0["pos"].copy(), g0["data_term"].copy()
    next_id = g0["V"]
    params = flame_amd.Params()
    gs = np.float32(1.25)
    ref = sync_oracle.RefGraph.from_flat(g0, feat_id)
    with flame_amd.Regularizer(0) as reg:
        reg.upload_graph(g0)
        kw = _mode(reg, "device" if mode == "prepared" else mode)
        tris = synth.delaunay_native(pos)[0]
        for frame in range(4):
            ids_now = feat_id
            reg.run(params, 30)
            # the dense map of this frame: two halves, the solver iterating in between
            reg.interpolate_mesh_begin(tris, h, w, graph_scale=float(gs))
            reg.run_async(params, 12)
            dense, cov = reg.interpolate_mesh_end()
            flat = sync_oracle.flatten(ref, ids_now)
            assert oracle.run(flat, 30) == 0
            assert oracle.run(flat, 12) == 0
            sync_oracle.absorb(ref, flat, ids_now)
            if frame == 0:  # the two-halves form against the one-call form on the same state
                reg.sync()
                chk = flame_amd.Regularizer(0)
                chk.upload_graph(g0)
                chk.run(params, 30)
                one, cov1 = chk.interpolate_mesh(tris, h, w, graph_scale=float(gs))
                chk.close()
                assert np.array_equal(one, dense, equal_nan=True) and cov1 == cov
            # next frame; the predictions the host would gather from the map
            feat_id, pos, data, next_id = next_frame(rng, feat_id, pos, data, next_id, w, h)
            weight = np.ones(len(feat_id), np.float32)
            edges = synth.delaunay_edges_scipy(pos)
            iy = (pos[:, 1] + np.float32(0.5)).astype(np.int32)
            ix = (pos[:, 0] + np.float32(0.5)).astype(np.int32)
            inside = (iy >= 0) & (iy < h) & (ix >= 0) & (ix < w)
            init_x = np.full(len(feat_id), np.nan, np.float32)
            init_x[inside] = dense[iy[inside], ix[inside]] / gs
            assert np.isnan(init_x).any() and (~np.isnan(init_x)).any()
            if mode == "prepared":
                reg.sync_prepare(feat_id, pos, data, weight, edges, init_graph_scale=float(gs), init_from_map=True, **kw)
                reg.run_async(params, 9)
                flat = sync_oracle.flatten(ref, ids_now)
                assert oracle.run(flat, 9) == 0
                sync_oracle.absorb(ref, flat, ids_now)
                reg.sync_commit()
            else:
                reg.sync_graph(feat_id, pos, data, weight, edges, init_graph_scale=float(gs), init_from_map=True, **kw)
            sync_oracle.sync(ref, feat_id, pos, data, weight, edges, init_x=init_x, init_graph_scale=float(gs))
            assert_state_equal(reg.download_state(), sync_oracle.flatten(ref, feat_id), keys=OUT_KEYS + ("x_prev",), what=f"frame {frame}")
            tris = synth.delaunay_native(pos)[0]
        with pytest.raises(flame_amd.NLTGV2Error):  # init_from_map needs the scale and excludes init_x
            reg.sync_graph(feat_id, pos, data, weight, edges, init_from_map=True)


def test_mesh_of_the_state_the_last_settle_left_beside_runs_in_flight(built):
    """FLAME_NLTGV2_OPT_MESH_STATE = 1: with runs enqueued since the last call that settled the solver, interpolate_mesh_begin rasterises
    the state THAT call left (the canonical arrays, which a run does not touch) and does not wait for the runs; 0 (default): it settles
    them and shows what they leave.  Both against the one-call form on a second context; after a sync_commit (the frame loop's use:
    commit, next round out, then the mesh) the map is the committed graph's, the state then moves on by exactly the enqueued runs."""
    import torch  # noqa: F401

    import flame_amd
    from flame_amd.regularizer import OPT_MESH_STATE

    w, h = 320, 240
    rng = np.random.default_rng(5)
    g0 = synth.make_graph("320x240", seed=15)
    feat_id = np.arange(g0["V"], dtype=np.int32)
    params = flame_amd.Params()
    tris = synth.delaunay_native(g0["pos"])[0]

    def one_call(n):
        with flame_amd.Regularizer(0) as chk:
            chk.upload_graph(g0)
            chk.run(params, n)
            return chk.interpolate_mesh(tris, h, w, graph_scale=1.5)

    with flame_amd.Regularizer(0) as reg:
        with pytest.raises(flame_amd.NLTGV2Error):
            reg.set_option(OPT_MESH_STATE, 2)
        reg.upload_graph(g0)
        reg.run(params, 40)
        reg.download_state()  # (a call that reads the state: the canonical arrays hold iteration 40 from here on)
        reg.set_option(OPT_MESH_STATE, 1)
        reg.run_async(params, 25)
        reg.run_async(params, 25)
        reg.interpolate_mesh_begin(tris, h, w, graph_scale=1.5)
        assert reg.runs_in_flight() >= 0  # (a query, not a wait)
        at40, cov40 = reg.interpolate_mesh_end()
        ref40, rcov40 = one_call(40)
        assert np.array_equal(at40, ref40, equal_nan=True) and cov40 == rcov40
        # the default waits for the runs and shows iteration 90
        reg.set_option(OPT_MESH_STATE, 0)
        reg.interpolate_mesh_begin(tris, h, w, graph_scale=1.5)
        at90, cov90 = reg.interpolate_mesh_end()
        ref90, rcov90 = one_call(90)
        assert np.array_equal(at90, ref90, equal_nan=True) and cov90 == rcov90
        assert not np.array_equal(at40, at90, equal_nan=True)
        # with the state canonical and nothing enqueued the option changes nothing
        reg.set_option(OPT_MESH_STATE, 1)
        reg.interpolate_mesh_begin(tris, h, w, graph_scale=1.5)
        again, _ = reg.interpolate_mesh_end()
        assert np.array_equal(again, at90, equal_nan=True)
        # the frame loop's order: prepare, iterate, commit, next round out, mesh of the committed graph
        ref = sync_oracle.RefGraph.from_flat(g0, feat_id)
        flat = sync_oracle.flatten(ref, feat_id)
        assert oracle.run(flat, 90) == 0
        sync_oracle.absorb(ref, flat, feat_id)
        ids_now = feat_id
        feat_id2, pos2, data2, _ = next_frame(rng, feat_id, g0["pos"].copy(), g0["data_term"].copy(), g0["V"], w, h)
        weight2 = np.ones(len(feat_id2), np.float32)
        edges2 = synth.delaunay_edges_scipy(pos2)
        tris2 = synth.delaunay_native(pos2)[0]
        reg.sync_prepare(feat_id2, pos2, data2, weight2, edges2)
        reg.run_async(params, 11)
        flat = sync_oracle.flatten(ref, ids_now)
        assert oracle.run(flat, 11) == 0
        sync_oracle.absorb(ref, flat, ids_now)
        reg.sync_commit()
        reg.run_async(params, 30)  # (the round a frame loop enqueues before it prepares the mesh)
        reg.interpolate_mesh_begin(tris2, h, w, graph_scale=1.5)
        committed, ccov = reg.interpolate_mesh_end()
        sync_oracle.sync(ref, feat_id2, pos2, data2, weight2, edges2)
        at_commit = sync_oracle.flatten(ref, feat_id2)
        with flame_amd.Regularizer(0) as chk:
            chk.upload_graph(at_commit)
            want, wcov = chk.interpolate_mesh(tris2, h, w, graph_scale=1.5)
        assert np.array_equal(committed, want, equal_nan=True) and ccov == wcov
        assert oracle.run(at_commit, 30) == 0
        assert_state_equal(reg.download_state(), at_commit, keys=OUT_KEYS, what="30 iterations behind the commit")


def test_commit_and_projection_behind_rounds_that_expire(built):
    """What a frame loop's holds enqueue behind rounds still in flight -- the next topology's expansion into the spare tables (beside the
    solver), the unpack of the state (behind the runs, before the host has seen how they ended) -- with rounds that EXPIRE
    (FLAME_NLTGV2_OPT_FAULT_INJECT: every wait gives up): the chain is redone on the per-step path, the state unpacked once more, the
    commit / the projection work on what the oracle has after the same iterations.  Three frames; then the same with healthy rounds."""
    import torch  # noqa: F401

    import flame_amd
    from flame_amd.regularizer import OPT_FAULT_INJECT

    w, h = 320, 240
    rng = np.random.default_rng(77)
    g0 = synth.make_graph("320x240", seed=16)
    feat_id = np.arange(g0["V"], dtype=np.int32)
    pos, data = g0["pos"].copy(), g0["data_term"].copy()
    next_id = g0["V"]
    params = flame_amd.Params()
    ref = sync_oracle.RefGraph.from_flat(g0, feat_id)
    with flame_amd.Regularizer(0) as reg:
        reg.upload_graph(g0)
        reg.set_feature_ids(feat_id)
        for frame in range(6):
            faulty, faulty_read = frame < 3, frame == 3  # (an expired run sends its topology to the per-step path for a few runs: one fault per topology)
            ids_now = feat_id
            feat_id, pos, data, next_id = next_frame(rng, feat_id, pos, data, next_id, w, h)
            weight = np.ones(len(feat_id), np.float32)
            edges = synth.delaunay_edges_scipy(pos)
            reg.sync_prepare(feat_id, pos, data, weight, edges)
            before = reg.info()["timeouts_recovered"]
            if faulty:
                reg.set_option(OPT_FAULT_INJECT, 200)
            reg.run_async(params, 30)
            reg.run_async(params, 20)
            reg.sync_commit()  # (the expansion goes out, the runs are found expired and redone, then the swap)
            if faulty:
                assert reg.info()["timeouts_recovered"] > before
                reg.set_option(OPT_FAULT_INJECT, 0)
            flat = sync_oracle.flatten(ref, ids_now)
            assert oracle.run(flat, 50) == 0
            sync_oracle.absorb(ref, flat, ids_now)
            sync_oracle.sync(ref, feat_id, pos, data, weight, edges)
            flat = sync_oracle.flatten(ref, feat_id)
            # rounds on the new graph, a call that reads the state behind them
            if faulty_read:
                reg.set_option(OPT_FAULT_INJECT, 200)
            before = reg.info()["timeouts_recovered"]
            reg.run_async(params, 25)
            got = reg.download_state()
            if faulty_read:
                assert reg.info()["timeouts_recovered"] > before
                reg.set_option(OPT_FAULT_INJECT, 0)
            assert oracle.run(flat, 25) == 0
            sync_oracle.absorb(ref, flat, feat_id)
            assert_state_equal(got, flat, keys=OUT_KEYS + ("x_prev",), what=f"frame {frame} ({'expired' if faulty else 'healthy'} rounds)")
            src, dst, fid = reg.topology()
            assert np.array_equal(src, flat["src"]) and np.array_equal(dst, flat["dst"]) and np.array_equal(fid, feat_id)


def test_vertex_counts_at_the_walk_padding_boundary(built):
    """Advisor, round 4: the device builder sized the per-vertex walk tables to V rounded up to 16384 bytes, topology_buffers then asked
    for V + 16 -- a reused buffer of exactly 16384 bytes was re-allocated AFTER the build for V in 16369..16384 and the tables were
    lost.  A context that first holds a graph of ~10.8 k vertices (buffers of 16384 bytes), then graphs of 16384 and 16383 vertices,
    built on the device: layout word for word as the host builders', 20 steps bit-identical."""
    import torch  # noqa: F401

    import flame_amd

    with flame_amd.Regularizer(0) as reg:
        for w, h in ((104, 104), (128, 128), (127, 129), (128, 128)):
            name = f"{6 * w}x{6 * h}"
            synth.CONFIGS[name] = (6 * w, 6 * h, 6)
            g = synth.make_graph(name, seed=3)
            assert g["V"] == w * h
            ref = synth.copy_graph(g)
            reg.upload_graph(g)
            assert reg.layout_selftest() == 0, name
            reg.run(flame_amd.Params(), 20)
            oracle.run(ref, 20)
            assert_state_equal(reg.download_state(), ref, keys=OUT_KEYS, what=name)
