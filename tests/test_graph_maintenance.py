"""Per-frame graph maintenance on the device state (second half):
Flame::projectGraph (flame.cc:1888-1905) and the rescale_data block (flame.cc:328-351) against their
restatements in the checker (oracle/photometric_oracle.c, UNPINNED), bit for bit, and followed by solver
steps so that the packed layouts provably pick up the moved positions."""
import numpy as np
import pytest

from flame_amd import synth
from oracle import capi as oracle
from tests.helpers import OUT_KEYS, assert_state_equal
from tests.test_photometric import _rot_y

pytestmark = pytest.mark.gpu


def quat_wxyz_from_rot(R):
    w = np.sqrt(max(0.0, 1.0 + R[0, 0] + R[1, 1] + R[2, 2])) / 2.0
    x = (R[2, 1] - R[1, 2]) / (4.0 * w)
    y = (R[0, 2] - R[2, 0]) / (4.0 * w)
    z = (R[1, 0] - R[0, 1]) / (4.0 * w)
    return np.array([w, x, y, z], np.float32)


def test_project_then_rescale_then_solve(built):
    import torch  # noqa: F401

    import flame_amd

    g = synth.make_graph("640x480", seed=17)
    g["x"][5] = 0.0  # exercises the infinite-depth branch (maxDepthProjection)
    g["x_bar"][5] = 0.0
    K = np.array([[525.0, 0, 320.0], [0, 525.0, 240.0], [0, 0, 1]], np.float64)
    R = _rot_y(0.02)
    t = np.array([0.03, -0.01, 0.02])
    K32, Kinv32 = K.astype(np.float32), np.linalg.inv(K).astype(np.float32)
    KRKinv = (K32 @ R.astype(np.float32) @ Kinv32).astype(np.float32)
    q = quat_wxyz_from_rot(R)
    region = (8.0, 8.0, 640.0 - 16.0, 480.0 - 16.0)  # border = rescale_factor_max*win_size/2 + 1 (flame.cc:1880)
    scale = np.float32(1.2)
    params = flame_amd.Params()

    ref = synth.copy_graph(g)
    with flame_amd.Regularizer(0) as reg:
        reg.upload_graph(g)
        reg.run(params, 30)
        oracle.run(ref, 30)
        # ---- projectGraph
        keep, pos = reg.project_graph(K32, Kinv32, KRKinv, q, t.astype(np.float32), region, graph_scale=float(scale))
        rkeep = oracle.graph_project(ref["pos"], ref["x"], float(scale), K32, Kinv32, q, t.astype(np.float32), KRKinv, region)
        assert np.array_equal(keep, rkeep) and 0 < keep.sum() < g["V"]
        assert np.array_equal(pos, ref["pos"])
        assert_state_equal(reg.download_state(), ref, what="after projectGraph")
        # the solver keeps running between projectGraph and syncGraph in the reference (separate lock
        # sections, flame.cc:302-318): positions moved, alpha not yet -> packed dx/dy must follow
        reg.run(params, 12)
        oracle.run(ref, 12)
        assert_state_equal(reg.download_state(), ref, what="steps after projectGraph")
        # ---- rescale_data
        new_scale = reg.rescale_data(float(scale), params)
        rs, rdf = oracle.graph_rescale(ref, float(scale), 0.1)
        assert np.float32(new_scale) == np.float32(rs) and np.float32(params.data_factor) == np.float32(rdf)
        st = reg.download_state()
        assert_state_equal(st, ref, keys=OUT_KEYS + ("x_prev",), what="after rescale")
        ref_p = oracle.make_params(data_factor=rdf)
        reg.run(params, 15)
        oracle.run(ref, 15, ref_p)
        assert_state_equal(reg.download_state(), ref, what="steps after rescale")


def test_projection_behind_rounds_in_flight_healthy_and_expired(built):
    """flame_nltgv2_project_graph with rounds still in flight: its kernel goes out behind them and the unpack of their state, before the
    host has seen how they ended.  Healthy rounds: one wait, the result of the oracle's projection after the same iterations.  Rounds
    that EXPIRE (FLAME_NLTGV2_OPT_FAULT_INJECT): the projection worked on rubbish -- the chain is redone, the state unpacked again, the
    positions come back from where the kernel kept them and it runs once more.  Two projections per case: the first keeps the old
    positions as the layout's (layout_pos), the second in pos_undo.  Solver steps in between show the packed records follow."""
    import torch  # noqa: F401

    import flame_amd
    from flame_amd.regularizer import OPT_FAULT_INJECT

    K = np.array([[525.0, 0, 320.0], [0, 525.0, 240.0], [0, 0, 1]], np.float64)
    K32, Kinv32 = K.astype(np.float32), np.linalg.inv(K).astype(np.float32)
    region = (8.0, 8.0, 640.0 - 16.0, 480.0 - 16.0)
    scale = np.float32(1.1)
    params = flame_amd.Params()
    for faulty in (False, True):
        g = synth.make_graph("640x480", seed=23)
        ref = synth.copy_graph(g)
        with flame_amd.Regularizer(0) as reg:
            reg.upload_graph(g)
            reg.run(params, 20)
            oracle.run(ref, 20)
            for k, angle in enumerate((0.015, -0.01)):
                R = _rot_y(angle)
                t = np.array([0.02, -0.01, 0.01], np.float32)
                KRKinv = (K32 @ R.astype(np.float32) @ Kinv32).astype(np.float32)
                q = quat_wxyz_from_rot(R)
                before = reg.info()["timeouts_recovered"]
                if faulty and k == 0:
                    reg.set_option(OPT_FAULT_INJECT, 200)
                reg.run_async(params, 17)
                reg.run_async(params, 9)
                keep, pos = reg.project_graph(K32, Kinv32, KRKinv, q, t, region, graph_scale=float(scale))
                if faulty and k == 0:
                    assert reg.info()["timeouts_recovered"] > before
                    reg.set_option(OPT_FAULT_INJECT, 0)
                oracle.run(ref, 26)
                rkeep = oracle.graph_project(ref["pos"], ref["x"], float(scale), K32, Kinv32, q, t, KRKinv, region)
                assert np.array_equal(keep, rkeep), (faulty, k)
                assert np.array_equal(pos, ref["pos"]), (faulty, k)
                assert_state_equal(reg.download_state(), ref, what=f"after projection {k} (expired rounds: {faulty})")
                reg.run(params, 8)
                oracle.run(ref, 8)
                assert_state_equal(reg.download_state(), ref, what=f"steps after projection {k} (expired rounds: {faulty})")
