"""GPU parity tests proper: the HIP path, called through the C-ABI (flame_amd.Regularizer is a thin
ctypes binding of include/flame_nltgv2.h), against the CPU checker in oracle/ on the same inputs.

Tolerance stated by north_star: converged inverse depth within 1e-4 RMS of the reference CPU solver.
The implementation is designed to be BIT-IDENTICAL (same IEEE operations in the same order, see
flame_amd/csrc/nltgv2_kernels.hip, nltgv2_persistent*.hip), so most tests assert exact equality; `TOL_RMS` is asserted as
well where the comparison is the north_star one.
"""
import numpy as np
import pytest

from flame_amd import synth
from tests.helpers import OUT_KEYS, assert_state_equal, load_golden, random_graph, rms

pytestmark = pytest.mark.gpu

TOL_RMS = 1e-4  # north_star: "<= 1e-4 RMS" on converged inverse depth


@pytest.fixture(scope="module")
def env(built):
    import torch  # noqa: F401  (first: one HIP runtime)

    import flame_amd
    from oracle import capi as oracle

    return flame_amd, oracle


def gpu_run(flame_amd, g, n, params=None, options=(), expect_path=None):
    with flame_amd.Regularizer(0) as reg:
        for k, v in options:
            reg.set_option(k, v)
        reg.upload_graph(g)
        reg.run(params or flame_amd.Params(), n)
        if expect_path is not None:
            assert reg.info()["last_run_path"] == expect_path
        return reg.download_state()


def cpu_run(oracle, g, n, **pkw):
    ref = synth.copy_graph(g)
    bad = oracle.run(ref, n, oracle.make_params(**pkw))
    return ref, bad


# ---- committed golden fixtures (reference-Triangle edge lists; outputs frozen in the build container)
@pytest.mark.parametrize("name", ["cfg1_320x240_s1234", "cfg1_320x240_s77_varied", "cfg2_640x480_s1234"])
def test_golden_fixture(env, name):
    flame_amd, _ = env
    g, z = load_golden(name)
    iters = [int(n) for n in z["iters"]]
    with flame_amd.Regularizer(0) as reg:
        reg.upload_graph(g)
        done = 0
        for n in iters:
            reg.run(flame_amd.Params(), n - done)
            done = n
            out = reg.download_state()
            for k in OUT_KEYS:
                key = f"n{n}_{k}"
                if key in z.files:
                    assert np.array_equal(out[k], z[key]), f"{name}: {key} differs"
            assert rms(out["x"], z[f"n{n}_x"]) <= TOL_RMS
            sm, dc = reg.costs(flame_amd.Params())
            # the reference's sequential float sums, to the last bit (the device forms the addends, the host adds them in order)
            assert [np.float32(sm), np.float32(dc)] == [np.float32(v) for v in z[f"n{n}_cost"]], (sm, dc, z[f"n{n}_cost"])


# ---- BASELINE.json configs at full size against the checker run on the same seeded inputs ----------
@pytest.mark.parametrize("config,n", [("320x240", 50), ("640x480", 200), ("1280x720", 200), ("1920x1080", 200)])
def test_config_parity(env, config, n):
    flame_amd, oracle = env
    g = synth.make_graph(config, seed=4242)
    ref, bad = cpu_run(oracle, g, n)
    assert bad == 0
    for persistent in (1, 3, 4, 6, 0):  # auto, vertex-per-lane, patch-per-wave (one / two half-edges per lane), one launch per step
        out = gpu_run(flame_amd, g, n, options=[(5, persistent)], expect_path=None if persistent else 2)
        assert rms(out["x"], ref["x"]) <= TOL_RMS
        assert_state_equal(out, ref, keys=OUT_KEYS + ("x_prev", "w1_prev", "w2_prev"), what=f"{config} p={persistent}")


@pytest.mark.parametrize("idx", [0, 1])
def test_config_hashes_fixture_on_gpu(env, idx):
    """tests/golden/config_hashes.json: the 1280x720 / 1920x1080 BASELINE graphs after 200 steps must hash (SHA-256 of
    all nine state arrays, both costs) to what the checker produced in the build container."""
    import hashlib
    import json
    import os

    from tests.conftest import ROOT

    flame_amd, _ = env
    e = json.load(open(os.path.join(ROOT, "tests", "golden", "config_hashes.json")))[idx]
    g = synth.make_graph(e["config"], e["seed"])
    sha = lambda a: hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()
    assert {k: sha(g[k]) for k in e["inputs"]} == e["inputs"], "the synthetic input itself differs on this machine"
    with flame_amd.Regularizer(0) as reg:
        reg.upload_graph(g)
        reg.run(flame_amd.Params(), e["iters"])
        out = reg.download_state()
        sm, dc = reg.costs(flame_amd.Params())
    assert {k: sha(out[k]) for k in e["state"]} == e["state"]
    assert [int(np.float32(sm).view(np.uint32)), int(np.float32(dc).view(np.uint32))] == e["costs_f32_bits"]


def test_four_kernel_path_matches_fused_and_checker(env):
    flame_amd, oracle = env
    g = synth.make_graph("320x240", seed=11)
    ref, _ = cpu_run(oracle, g, 37)
    persistent = gpu_run(flame_amd, g, 37, expect_path=6)  # auto: the patch-per-wave form
    persistent_tv = gpu_run(flame_amd, g, 37, options=[(5, 3)], expect_path=5)
    assert_state_equal(persistent_tv, ref, what="persistent vertex-per-lane")
    with flame_amd.Regularizer(0) as reg:  # (2 named round 1's lane-per-half-edge form, retired in round 3)
        with pytest.raises(flame_amd.NLTGV2Error):
            reg.set_option(5, 2)
    fused = gpu_run(flame_amd, g, 37, options=[(5, 0)], expect_path=2)
    canon = gpu_run(flame_amd, g, 37, options=[(flame_amd.regularizer.OPT_SOLVER, 1)], expect_path=4)
    assert_state_equal(persistent, ref, what="persistent")
    assert_state_equal(fused, ref, what="fused")
    assert_state_equal(canon, ref, what="4-kernel")


def test_internal_steps_individually(env):
    """internal::dualStep / primalStep / extraGradientStep (h:158-168) one at a time."""
    flame_amd, oracle = env
    g = synth.make_graph("320x240", seed=5)
    ref = synth.copy_graph(g)
    oracle.run(ref, 3)  # non-trivial state
    p = flame_amd.Params()
    with flame_amd.Regularizer(0) as reg:
        reg.upload_graph(ref)
        for _ in range(2):
            ref["x_prev"][:] = ref["x"]; ref["w1_prev"][:] = ref["w1"]; ref["w2_prev"][:] = ref["w2"]
            reg.save_prev()
            assert oracle.dual_step(ref) == 0
            reg.dual_step(p)
            assert_state_equal(reg.download_state(), ref, what="dual")
            oracle.primal_step(ref)
            reg.primal_step(p)
            assert_state_equal(reg.download_state(), ref, what="primal")
            oracle.extragradient_step(ref)
            reg.extragradient_step(p)
            assert_state_equal(reg.download_state(), ref, keys=OUT_KEYS + ("x_prev",), what="extragradient")
        # and mixing granular sweeps with fused runs keeps one consistent state
        reg.run(p, 5)
        oracle.run(ref, 5)
        assert_state_equal(reg.download_state(), ref, what="mixed")


def _opt_sets():
    from flame_amd.regularizer import (OPT_BLOCK_WAVES as BW, OPT_DUAL_PUBLISH as DUAL, OPT_PERSISTENT as P, OPT_PLACEMENT as PLACE,
                                       OPT_POLL_GAP as GAP, OPT_PRESLEEP as PRE, OPT_PROBE as PROBE, OPT_UNROLL as U,
                                       OPT_USE_HIPGRAPH as HG, OPT_VERIFY_RECORDS as VERIFY, OPT_XCDS as XCDS)
    return [
        # one launch per step: waves per workgroup x slot chunk, hipGraph off
        [(P, 0), (BW, 1), (U, 4)], [(P, 0), (BW, 1), (U, 16)], [(P, 0), (BW, 4), (U, 8)], [(P, 0), (HG, 0)],
        [(P, 1)],  # automatic choice
        # vertex per lane
        [(P, 3)], [(P, 3), (DUAL, 0)], [(P, 3), (XCDS, 1)], [(P, 3), (XCDS, 4)], [(P, 3), (PRE, 21)],
        # patch per wave
        [(P, 4)], [(P, 4), (DUAL, 0)], [(P, 4), (DUAL, 2)], [(P, 4), (XCDS, 1)], [(P, 4), (XCDS, 8)], [(P, 4), (GAP, 1)], [(P, 4), (GAP, 2)],
        [(P, 4), (GAP, 4)], [(P, 4), (PRE, 9), (GAP, 4)], [(P, 4), (PROBE, 1)], [(P, 4), (PLACE, 0)], [(P, 4), (PLACE, 0), (DUAL, 0)],
        [(P, 4), (PLACE, 1), (VERIFY, 1)],
        # patch per wave, two half-edges per lane
        [(P, 6)], [(P, 6), (DUAL, 0)], [(P, 6), (GAP, 4)], [(P, 6), (PRE, 9), (GAP, 4)], [(P, 6), (VERIFY, 1)],
    ]


@pytest.mark.parametrize("opts", _opt_sets(), ids=lambda o: "-".join(f"{k}={v}" for k, v in o))
def test_launch_configurations_are_bit_identical(env, opts):
    """Every launch configuration computes the same bits: the one-launch-per-step sweep (waves per workgroup, slot chunk,
    hipGraph on / off), the automatic choice and the persistent forms with their knobs -- same-XCD exchange through L2
    on / off, slot constants in LDS, the pre-poll pause, the XCDs a launch is spread over, the poll pacing, the cycle probe,
    the placement of the records read across XCDs and the record verification."""
    flame_amd, oracle = env
    g = synth.make_graph("320x240", seed=3)
    ref, _ = cpu_run(oracle, g, 21)
    assert_state_equal(gpu_run(flame_amd, g, 21, options=opts), ref, what=str(opts))


def test_step_by_step_equals_one_run(env):
    flame_amd, oracle = env
    g = synth.make_graph("320x240", seed=8)
    ref, _ = cpu_run(oracle, g, 9)
    with flame_amd.Regularizer(0) as reg:
        reg.upload_graph(g)
        for _ in range(9):  # odd count: exercises both ping-pong parities and the non-graph path
            reg.step(flame_amd.Params())
        assert_state_equal(reg.download_state(), ref, what="9 x step")
        reg.run(flame_amd.Params(), 300)  # > one hipGraph chunk (256) + remainder
        oracle.run(ref, 300)
        assert_state_equal(reg.download_state(), ref, what="309")


def test_non_default_params_and_weights(env):
    flame_amd, oracle = env
    g = random_graph(700, 2500, seed=1)
    kw = dict(data_factor=0.37, step_x=0.004, step_q=31.0, theta=0.6, x_min=0.7, x_max=1.3)
    ref, bad = cpu_run(oracle, g, 60, **kw)
    assert bad == 0
    out = gpu_run(flame_amd, g, 60, params=flame_amd.Params(**kw))
    assert_state_equal(out, ref, what="params")
    assert out["x"].min() >= np.float32(0.7) and out["x"].max() <= np.float32(1.3)


# ---- edge cases ------------------------------------------------------------------------------------
def test_empty_and_tiny_graphs(env):
    flame_amd, oracle = env
    p = flame_amd.Params()
    cases = [
        synth.assemble_graph(np.zeros((0, 2), np.float32), np.zeros(0, np.float32), np.zeros((0, 2), np.int32)),
        synth.assemble_graph(np.array([[1, 2]], np.float32), np.array([0.9], np.float32), np.zeros((0, 2), np.int32)),
        synth.assemble_graph(np.array([[0, 0], [3, 4]], np.float32), np.array([0.5, 1.5], np.float32),
                             np.array([[1, 0]], np.int32)),
        random_graph(65, 0, seed=2),    # isolated vertices only, two slices
        random_graph(64, 200, seed=3),  # exactly one full slice
        random_graph(129, 400, seed=4),  # ragged last slice
    ]
    for g in cases:
        ref, _ = cpu_run(oracle, g, 13)
        with flame_amd.Regularizer(0) as reg:
            reg.upload_graph(g)
            reg.run(p, 13)
            out = reg.download_state()
            assert_state_equal(out, ref, what=f"V={g['V']} E={g['E']}")
            sm, dc = reg.costs(p)
            rs, rd = oracle.costs(ref)
            assert np.float32(sm) == np.float32(rs) and np.float32(dc) == np.float32(rd), (sm, rs, dc, rd)


def test_star_graph_high_degree(env):
    """One hub with degree 999 (slice width 999) + chain; both orientations."""
    flame_amd, oracle = env
    V = 1000
    rng = np.random.default_rng(0)
    pos = (rng.random((V, 2)) * 50).astype(np.float32)
    data = (0.5 + rng.random(V)).astype(np.float32)
    e = [(0, i) if i % 2 else (i, 0) for i in range(1, V)] + [(i, i + 1) for i in range(1, V - 1)]
    g = synth.assemble_graph(pos, data, np.array(e, np.int32))
    ref, bad = cpu_run(oracle, g, 25)
    assert bad == 0
    for unroll in (4, 16):  # one launch per step
        out = gpu_run(flame_amd, g, 25, options=[(5, 0), (flame_amd.regularizer.OPT_UNROLL, unroll)], expect_path=2)
        assert_state_equal(out, ref, what=f"star U={unroll}")
    # degree 999 > 64 lanes: no patch can hold it, and the vertex-per-lane form cannot either (999 > 8*64) -> automatic
    # fall back to per-step launches
    out = gpu_run(flame_amd, g, 25, expect_path=2)
    assert_state_equal(out, ref, what="star auto")
    # a hub of degree 300: 38 chained lanes in the vertex-per-lane form
    V2 = 400
    e2 = [(0, i) if i % 3 else (i, 0) for i in range(1, 301)] + [(i, i + 1) for i in range(1, V2 - 1)]
    g2 = synth.assemble_graph(pos[:V2], data[:V2], np.array(e2, np.int32))
    ref2, _ = cpu_run(oracle, g2, 25)
    for form, path in ((3, 5), (1, 5)):  # forced, and picked automatically (degree 300 > 64 lanes)
        out = gpu_run(flame_amd, g2, 25, options=[(5, form)], expect_path=path)
        assert_state_equal(out, ref2, what=f"hub of degree 300, persistent option {form}")


@pytest.mark.parametrize("hub_degrees", [(17,), (18, 31, 32), (33, 47, 48), (49, 63, 64), (16, 17, 20, 40, 64)])
def test_vertices_of_more_than_sixteen_edges_in_the_patch_kernel(env, hub_degrees):
    """k_persistent_pv adds a vertex's contributions up across the lanes of a 16-lane row; a vertex of 17..64 edges fills two
    to four rows of its patch and its sum runs row after row (the running sums handed from the first lane of one row to the
    next).  Hubs at every row boundary, both edge orientations, odd and even run lengths: both persistent forms and the
    per-step path agree with the checker bit for bit."""
    flame_amd, oracle = env
    rng = np.random.default_rng(sum(hub_degrees))
    g0 = synth.make_graph("320x240", seed=40 + len(hub_degrees))
    V = g0["V"]
    edges = [tuple(e) for e in np.stack([g0["src"], g0["dst"]], 1)]
    deg = np.bincount(np.concatenate([g0["src"], g0["dst"]]), minlength=V)
    hubs = rng.choice(V, len(hub_degrees), replace=False)
    for h, want in zip(hubs, hub_degrees):
        have = {b if a == h else a for a, b in edges if h in (a, b)}
        cands = [v for v in rng.permutation(V) if v != h and v not in have and v not in hubs and deg[v] < 12]
        for v in cands[: max(0, want - len(have))]:
            edges.append((h, v) if rng.random() < 0.5 else (v, h))
            deg[v] += 1
        deg[h] = want
    perm = rng.permutation(len(edges))  # the hubs' edges spread over the edge list (ascending edge id = accumulation order)
    e = np.array(edges, np.int32)[perm]
    g = synth.assemble_graph(g0["pos"], g0["data_term"], e)
    got_deg = np.bincount(np.concatenate([g["src"], g["dst"]]), minlength=V)
    assert sorted(got_deg[hubs]) == sorted(hub_degrees), (got_deg[hubs], hub_degrees)
    for n in (1, 2, 37):
        ref, bad = cpu_run(oracle, g, n)
        assert bad == 0
        for form, path in ((4, 6), (3, 5)) + (((6, 7),) if max(hub_degrees) <= 32 else ()):
            out = gpu_run(flame_amd, g, n, options=[(5, form)], expect_path=path if n >= 4 else None)
            assert_state_equal(out, ref, what=f"hubs {hub_degrees}, form {form}, {n} steps")
    out = gpu_run(flame_amd, g, 37, options=[(5, 0)], expect_path=2)
    assert_state_equal(out, cpu_run(oracle, g, 37)[0], what=f"hubs {hub_degrees}, per step")


def test_edge_order_and_orientation_semantics(env):
    """Shuffling edge ORDER changes results only at rounding level; flipping ORIENTATION is a
    different operator (SURVEY.md 7 'Orientation is semantic').  GPU follows the checker in both."""
    flame_amd, oracle = env
    g = synth.make_graph("320x240", seed=21)
    base = gpu_run(flame_amd, g, 100)
    perm = np.random.default_rng(1).permutation(g["E"])
    gs = synth.copy_graph(g)
    for k in ("src", "dst", "alpha", "beta"):
        gs[k] = np.ascontiguousarray(g[k][perm])
    shuf = gpu_run(flame_amd, gs, 100)
    ref_s, _ = cpu_run(oracle, gs, 100)
    assert_state_equal(shuf, ref_s, keys=("x", "w1", "w2"), what="shuffled")
    assert rms(shuf["x"], base["x"]) < 2e-6
    gf = synth.copy_graph(g)
    gf["src"], gf["dst"] = g["dst"].copy(), g["src"].copy()
    flip = gpu_run(flame_amd, gf, 100)
    ref_f, _ = cpu_run(oracle, gf, 100)
    assert_state_equal(flip, ref_f, what="flipped")
    assert rms(flip["x"], base["x"]) > 1e-5


def test_batch_of_frames_equals_individual_frames(env):
    """A batch of independent frames is uploaded as a disjoint union: every frame's result must be
    what it is when solved alone (no cross-talk through the packed layout)."""
    flame_amd, oracle = env
    frames = [synth.make_graph("320x240", seed=100 + i) for i in range(5)]
    union = synth.concat_graphs(frames)
    refs = [cpu_run(oracle, f, 40)[0] for f in frames]
    for opts in ([], [(5, 4)], [(5, 3)]):  # auto; patch-per-wave form; vertex-per-lane form
        out = gpu_run(flame_amd, union, 40, options=opts)
        vo = eo = 0
        for f, ref in zip(frames, refs):
            for k in ("x", "w1", "w2", "x_bar"):
                assert np.array_equal(out[k][vo:vo + f["V"]], ref[k]), (opts, k)
            for k in ("q1", "q2", "q3"):
                assert np.array_equal(out[k][eo:eo + f["E"]], ref[k]), (opts, k)
            vo += f["V"]
            eo += f["E"]


def test_large_batch_as_groups_of_patches(env):
    """12 frames of 640x480 in the patch-per-wave form exceed its residency cap: groups of whole frames."""
    flame_amd, oracle = env
    frames = [synth.make_graph("640x480", seed=500 + i) for i in range(12)]
    union = synth.concat_graphs(frames)
    with flame_amd.Regularizer(0) as reg:
        reg.set_option(5, 4)
        reg.upload_graph(union)
        reg.run(flame_amd.Params(), 20)
        info = reg.info()
        out = reg.download_state(("x", "q3"))
    assert info["last_run_path"] == 6 and info["last_run_groups"] >= 2
    vo = eo = 0
    for i, f in enumerate(frames):
        if i % 5 == 0 or i == len(frames) - 1:
            ref, _ = cpu_run(oracle, f, 20)
            assert np.array_equal(out["x"][vo:vo + f["V"]], ref["x"]), i
            assert np.array_equal(out["q3"][eo:eo + f["E"]], ref["q3"]), i
        vo += f["V"]
        eo += f["E"]


def test_seven_frames_in_one_patch_per_wave_launch(env):
    """Seven frames of 640x480 would be 26 one-half-edge patches per CU; the planner runs them in ONE launch of the
    two-half-edges-per-lane form (13 waves per CU), and -- by name -- in one launch of the one-half-edge form (28 patches per CU are
    really resident: the kernel is capped at 92 SGPRs); every frame equals the frame solved alone."""
    flame_amd, oracle = env
    frames = [synth.make_graph("640x480", seed=700 + i) for i in range(7)]
    union = synth.concat_graphs(frames)
    refs = {i: cpu_run(oracle, frames[i], 33)[0] for i in (0, 3, 6)}
    for form, path in ((1, 7), (4, 6)):
        with flame_amd.Regularizer(0) as reg:
            reg.set_option(5, form)
            reg.upload_graph(union)
            reg.run(flame_amd.Params(), 33)
            info = reg.info()
            out = reg.download_state(("x", "w1", "x_bar", "q2"))
        assert info["last_run_path"] == path and info["last_run_groups"] == 1 and info["patches"] > 25 * 256, info
        vo = eo = 0
        for i, f in enumerate(frames):
            if i in refs:
                for k in ("x", "w1", "x_bar"):
                    assert np.array_equal(out[k][vo:vo + f["V"]], refs[i][k]), (form, i, k)
                assert np.array_equal(out["q2"][eo:eo + f["E"]], refs[i]["q2"]), (form, i)
            vo += f["V"]
            eo += f["E"]


def test_large_batch_runs_as_groups_of_resident_frames(env):
    """40 frames do not fit the chip at once: the persistent run is split into groups of whole frames
    (connected components), which is the same computation because frames are independent."""
    flame_amd, oracle = env
    frames = [synth.make_graph("640x480", seed=300 + i) for i in range(40)]
    union = synth.concat_graphs(frames)
    with flame_amd.Regularizer(0) as reg:
        reg.upload_graph(union)
        reg.run(flame_amd.Params(), 30)
        info = reg.info()
        out = reg.download_state(("x", "w2", "x_bar", "q1"))
    assert info["last_run_path"] == 5 and info["last_run_groups"] >= 2
    vo = eo = 0
    for i, f in enumerate(frames):
        if i % 7 == 0 or i == len(frames) - 1:
            ref, _ = cpu_run(oracle, f, 30)
            for k in ("x", "w2", "x_bar"):
                assert np.array_equal(out[k][vo:vo + f["V"]], ref[k]), (i, k)
            assert np.array_equal(out["q1"][eo:eo + f["E"]], ref["q1"]), i
        vo += f["V"]
        eo += f["E"]


def test_update_data_and_upload_state(env):
    """Per-frame path with unchanged topology: refresh data term (flam.cc:1985-2018), warm start kept;
    then a host-side rescale of x (flame.cc:328-351) pushed with upload_state."""
    flame_amd, oracle = env
    p = flame_amd.Params()
    g = synth.make_graph("320x240", seed=31)
    ref = synth.copy_graph(g)
    with flame_amd.Regularizer(0) as reg:
        reg.upload_graph(g)
        reg.run(p, 20)
        oracle.run(ref, 20)
        new_data = (ref["data_term"] * np.float32(1.01)).astype(np.float32)
        new_w = np.full(g["V"], 2.0, np.float32)
        ref["data_term"], ref["data_weight"] = new_data, new_w
        reg.update_data(new_data, new_w)
        reg.run(p, 20)
        oracle.run(ref, 20)
        assert_state_equal(reg.download_state(), ref, what="update_data")
        s = np.float32(1.25)
        for k in ("x", "x_bar", "x_prev"):
            ref[k] = (ref[k] / s).astype(np.float32)
        reg.upload_state({k: ref[k] for k in ("x", "x_bar", "x_prev")})
        reg.run(p, 7)
        oracle.run(ref, 7)
        assert_state_equal(reg.download_state(), ref, what="upload_state")


def test_reupload_grows_and_shrinks(env):
    flame_amd, oracle = env
    p = flame_amd.Params()
    with flame_amd.Regularizer(0) as reg:
        for cfg, seed in (("320x240", 1), ("640x480", 2), ("320x240", 3)):
            g = synth.make_graph(cfg, seed=seed)
            ref, _ = cpu_run(oracle, g, 30)
            reg.upload_graph(g)
            reg.run(p, 30)
            assert_state_equal(reg.download_state(), ref, what=cfg)


def test_nan_is_reported_not_fatal(env):
    """The reference exit(1)s through FLAME_ASSERT(!isnan(new_q)) (h:174); the library returns
    FLAME_NLTGV2_ERR_NAN and stays usable."""
    flame_amd, oracle = env
    g = synth.make_graph("320x240", seed=9)
    bad = synth.copy_graph(g)
    bad["x_bar"][17] = np.nan
    refbad = synth.copy_graph(bad)
    assert oracle.run(refbad, 1) != 0
    with flame_amd.Regularizer(0) as reg:
        reg.upload_graph(bad)
        with pytest.raises(flame_amd.NLTGV2Error) as ei:
            reg.run(flame_amd.Params(), 1)
        assert ei.value.status == flame_amd.regularizer.ERR_NAN
        # reported once, not sticky: the state stays readable (the caller can look at what happened) ...
        st = reg.download_state(("x", "q1"))
        assert st["x"].shape[0] == g["V"] and np.abs(st["q1"]).max() <= 1.0
        reg.costs(flame_amd.Params())
        for opt in (3, 4, 0):  # ... and every persistent form reports it the same way
            reg.set_option(5, opt)
            reg.upload_graph(bad)
            with pytest.raises(flame_amd.NLTGV2Error) as ei:
                reg.run(flame_amd.Params(), 6)
            assert ei.value.status == flame_amd.regularizer.ERR_NAN, opt
        reg.set_option(5, 1)
        reg.upload_graph(g)  # recovers
        reg.run(flame_amd.Params(), 2)
        ref, _ = cpu_run(oracle, g, 2)
        assert_state_equal(reg.download_state(), ref, what="after NaN")
    with flame_amd.Regularizer(0) as reg:  # same through the 4-kernel path
        reg.set_option(1, 1)
        reg.upload_graph(bad)
        with pytest.raises(flame_amd.NLTGV2Error):
            reg.run(flame_amd.Params(), 1)


@pytest.mark.parametrize("form", [3, 4, 6])
def test_persistent_timeout_is_rolled_back_and_redone(env, form):
    """A persistent run whose neighbour wait expires (fault injection: one wave withholds its first record) must
    leave the state it started from untouched; run() then does the same steps with one launch per step."""
    flame_amd, oracle = env
    g = synth.make_graph("320x240", seed=9)
    p = flame_amd.Params()
    ref = synth.copy_graph(g)
    with flame_amd.Regularizer(0) as reg:
        reg.set_option(5, form)
        reg.upload_graph(g)
        reg.run(p, 30)                       # a normal persistent run first (odd/even parity both follow)
        oracle.run(ref, 30)
        assert reg.info()["last_run_path"] in (5, 6, 7)
        reg.set_option(flame_amd.regularizer.OPT_FAULT_INJECT, 200)
        reg.run(p, 41)                       # times out inside, recovered
        oracle.run(ref, 41)
        info = reg.info()
        # (taken back, then redone in a persistent form at reduced residency, where the hook's fault is off)
        assert info["timeouts_recovered"] == 1 and info["last_run_path"] in (5, 6, 7)
        assert_state_equal(reg.download_state(), ref, keys=OUT_KEYS + ("x_prev", "w1_prev", "w2_prev"), what="after recovery")
        for _ in range(6):                   # the topology stays on the per-step path while the fault is on (and for a few runs after an
            reg.run(p, 10)                   # expired run in any case: 4, then 8, ... up to 1024 -- a stall that passes is tried again)
            oracle.run(ref, 10)
        assert reg.info()["timeouts_recovered"] == 1
        reg.set_option(flame_amd.regularizer.OPT_FAULT_INJECT, 0)  # fault off: persistent runs again
        reg.run(p, 25)
        oracle.run(ref, 25)
        assert reg.info()["last_run_path"] in (5, 6, 7)
        assert_state_equal(reg.download_state(), ref, keys=OUT_KEYS + ("x_prev", "w1_prev", "w2_prev"), what="after the fault")
        # chained asynchronous runs (run_async back to back, an asynchronous export in between, a short per-step run):
        # the chain's starting state was copied aside, the whole chain is replayed on the per-step path
        import torch

        reg.set_option(flame_amd.regularizer.OPT_FAULT_INJECT, 200)
        buf = torch.zeros(g["V"], dtype=torch.float32, device="cuda")
        before = reg.info()["timeouts_recovered"]
        reg.run_async(p, 8)
        reg.run_async(p, 9)
        reg.export_idepth_device(buf.data_ptr(), 3.0, wait=False)
        reg.run_async(p, 2)
        reg.run_async(p, 12)
        reg.sync()
        oracle.run(ref, 17)
        want_export = ref["x"] * np.float32(3.0)
        oracle.run(ref, 14)
        assert reg.info()["timeouts_recovered"] == before + 1
        assert_state_equal(reg.download_state(), ref, keys=OUT_KEYS + ("x_prev", "w1_prev", "w2_prev"), what="after a replayed chain")
        torch.cuda.synchronize()
        assert np.array_equal(buf.cpu().numpy(), want_export)
        # and a chain that does not time out is left alone
        reg.set_option(flame_amd.regularizer.OPT_FAULT_INJECT, 0)
        reg.run_async(p, 6)
        reg.run_async(p, 7)
        reg.sync()
        oracle.run(ref, 13)
        assert reg.info()["timeouts_recovered"] == before + 1
        assert_state_equal(reg.download_state(), ref, keys=OUT_KEYS + ("x_prev", "w1_prev", "w2_prev"), what="after a clean chain")


@pytest.mark.parametrize("form", [3, 4, 6])
def test_record_verification_detects_a_corrupted_read_and_recovers(env, form):
    """FLAME_NLTGV2_OPT_VERIFY_RECORDS: the persistent kernels re-read every neighbour record after its tag matched and
    compare all four dwords -- the run-time guard of the one hardware property the exchange relies on (an aligned 16-byte
    access is never torn between payload and tag).  Clean runs pass it; with the test hook (one re-read gets a flipped
    payload bit) the run is stopped, counted, rolled back and redone with one launch per step: same result."""
    flame_amd, oracle = env
    g = synth.make_graph("320x240", seed=19)
    p = flame_amd.Params()
    ref = synth.copy_graph(g)
    with flame_amd.Regularizer(0) as reg:
        reg.set_option(5, form)
        reg.set_option(14, 1)
        reg.upload_graph(g)
        reg.run(p, 60)
        oracle.run(ref, 60)
        info = reg.info()
        assert info["torn_records_detected"] == 0 and info["timeouts_recovered"] == 0 and info["last_run_path"] in (5, 6, 7)
        assert_state_equal(reg.download_state(), ref, keys=OUT_KEYS + ("x_prev", "w1_prev", "w2_prev"), what="verified run")
        reg.set_option(14, 2)  # the hook corrupts one re-read in step 2 of the next persistent run
        reg.run(p, 33)
        oracle.run(ref, 33)
        info = reg.info()
        assert info["torn_records_detected"] == 1 and info["timeouts_recovered"] == 0 and info["last_run_path"] in (2, 3)
        assert_state_equal(reg.download_state(), ref, keys=OUT_KEYS + ("x_prev", "w1_prev", "w2_prev"), what="after a detected corruption")
        reg.set_option(14, 1)
        reg.run(p, 20)
        oracle.run(ref, 20)
        assert reg.info()["last_run_path"] in (5, 6, 7) and reg.info()["torn_records_detected"] == 1
        assert_state_equal(reg.download_state(), ref, keys=OUT_KEYS + ("x_prev", "w1_prev", "w2_prev"), what="verified again")


def test_invalid_arguments(env):
    flame_amd, _ = env
    g = synth.make_graph("320x240", seed=9)
    with flame_amd.Regularizer(0) as reg:
        with pytest.raises(flame_amd.NLTGV2Error) as ei:
            reg.run(flame_amd.Params(), 1)  # nothing uploaded
        assert ei.value.status == -4
        b = synth.copy_graph(g)
        b["dst"][5] = g["V"]  # out of range
        with pytest.raises(flame_amd.NLTGV2Error):
            reg.upload_graph(b)
        b = synth.copy_graph(g)
        b["dst"][5] = b["src"][5]  # self loop
        with pytest.raises(flame_amd.NLTGV2Error):
            reg.upload_graph(b)
    with pytest.raises(flame_amd.NLTGV2Error):
        flame_amd.Regularizer(99)


def test_long_run_no_drift(env):
    """5000 steps ('>= 5000 to show no drift'): still bit-identical, and converged
    (per-step change at the prox-snapping plateau)."""
    flame_amd, oracle = env
    g = synth.make_graph("320x240", seed=77)
    ref, bad = cpu_run(oracle, g, 5000)
    assert bad == 0
    with flame_amd.Regularizer(0) as reg:
        reg.upload_graph(g)
        reg.run(flame_amd.Params(), 5000)
        out = reg.download_state()
        assert_state_equal(out, ref, what="5000")
        reg.run(flame_amd.Params(), 1)
        nxt = reg.download_state(("x",))
    assert rms(nxt["x"], out["x"]) < 5e-5


def test_export_idepth_device_and_stream(env):
    """x * graph_scale written to a device buffer in the caller's vertex order (flame.cc:377), with
    the solver running on the host framework's stream."""
    import torch

    flame_amd, oracle = env
    g = synth.make_graph("320x240", seed=13)
    ref, _ = cpu_run(oracle, g, 10)
    with flame_amd.Regularizer(0) as reg:
        s = torch.cuda.Stream()
        reg.set_stream(s.cuda_stream)
        reg.upload_graph(g)
        reg.run(flame_amd.Params(), 10)
        buf = torch.empty(g["V"], dtype=torch.float32, device="cuda:0")
        reg.export_idepth_device(buf.data_ptr(), 2.5)
        torch.cuda.synchronize()
        assert np.array_equal(buf.cpu().numpy(), ref["x"] * np.float32(2.5))
        st = reg.download_state(("x",))  # canonical now current; export again from that form
        reg.export_idepth_device(buf.data_ptr(), 1.0)
        torch.cuda.synchronize()
        assert np.array_equal(buf.cpu().numpy(), st["x"])
        reg.set_stream(None)


@pytest.mark.parametrize("form", [0, 3, 4, 6])
def test_standing_export_target(env, form):
    """flame_nltgv2_set_export_target: every run leaves scale * x in the caller's vertex order, on all paths."""
    import torch

    flame_amd, oracle = env
    g = synth.make_graph("320x240", seed=21)
    ref = synth.copy_graph(g)
    p = flame_amd.Params()
    buf = torch.full((g["V"],), -7.0, dtype=torch.float32, device="cuda")
    with flame_amd.Regularizer(0) as reg:
        reg.set_option(5, form)
        reg.upload_graph(g)
        reg.set_export_target(buf.data_ptr(), 1.5)
        for n in (17, 4, 30):
            reg.run(p, n)
            oracle.run(ref, n)
            torch.cuda.synchronize()
            assert np.array_equal(buf.cpu().numpy(), ref["x"] * np.float32(1.5)), (form, n)
        reg.set_export_target(None)
        buf.fill_(-7.0)
        reg.run(p, 5)
        torch.cuda.synchronize()
        assert float(buf.max()) == -7.0


@pytest.mark.parametrize("form", [0, 1, 3, 4, 6])
def test_another_stream_waits_for_the_run_through_the_launch_own_event(env, form):
    """flame_nltgv2_stream_wait_run: a consumer stream of the caller's is ordered behind the runs enqueued so far -- with the event the
    run's own launch carries (called right behind run_async), or with one recorded on the spot (the first launch of a topology is
    cooperative; another call came in between; the per-step path).  The consumer copies the export rows of alternating targets (more
    than the context's four epilogue-argument slots) and must always see the finished row; the solver's stream is never waited for
    by the host in between."""
    import torch

    flame_amd, oracle = env
    g = synth.make_graph("320x240", seed=33)
    ref = synth.copy_graph(g)
    p = flame_amd.Params()
    n_rows = 6
    rows = torch.full((n_rows, g["V"]), -5.0, dtype=torch.float32, device="cuda")
    seen = torch.zeros((2 * n_rows, g["V"]), dtype=torch.float32, device="cuda")
    solver, side = torch.cuda.Stream(priority=-1), torch.cuda.Stream()
    want = []
    with flame_amd.Regularizer(0) as reg:
        reg.set_option(5, form)
        reg.set_stream(solver.cuda_stream)
        reg.upload_graph(g)
        with pytest.raises(flame_amd.NLTGV2Error):
            reg.stream_wait_run(solver.cuda_stream)  # (the context's own stream)
        with pytest.raises(flame_amd.NLTGV2Error):
            reg.stream_wait_run(0)
        for k in range(2 * n_rows):
            reg.set_export_target(rows[k % n_rows].data_ptr(), 1.0 + k)
            reg.run_async(p, 7 + k)
            if k % 3 == 2:
                reg.info()  # (a call between the run and the wait: the event is recorded instead)
            reg.stream_wait_run(side.cuda_stream)
            with torch.cuda.stream(side):
                seen[k].copy_(rows[k % n_rows], non_blocking=True)
            oracle.run(ref, 7 + k)
            want.append(ref["x"] * np.float32(1.0 + k))
        side.synchronize()
        got = seen.cpu().numpy()
        for k in range(2 * n_rows):
            assert np.array_equal(got[k], want[k]), (form, k)
        reg.sync()
        assert reg.info()["timeouts_recovered"] == 0


@pytest.mark.parametrize("form", [0, 1, 6])
def test_runs_in_flight_counts_the_last_two_runs_without_waiting(env, form):
    """flame_nltgv2_runs_in_flight: 0 on an idle context, at most 2, falls back to 0 once the device has finished the runs (no sync by
    the caller), and the runs it watched are as good as any: the state equals the checker's."""
    import time

    flame_amd, oracle = env
    g = synth.make_graph("640x480", seed=41)
    ref = synth.copy_graph(g)
    p = flame_amd.Params()
    with flame_amd.Regularizer(0) as reg:
        reg.set_option(5, form)
        reg.upload_graph(g)
        assert reg.runs_in_flight() == 0
        seen, total = set(), 0
        for k in range(6):
            reg.run_async(p, 400)
            total += 400
            n = reg.runs_in_flight()
            assert 0 <= n <= 2
            seen.add(n)
        assert max(seen) >= 1  # (2 400 iterations take > 2 ms: the query returned long before)
        t0 = time.perf_counter()
        while reg.runs_in_flight() != 0:
            assert time.perf_counter() - t0 < 10.0
            time.sleep(0.0005)
        reg.sync()
        oracle.run(ref, total)
        out = reg.download_state()
        assert all(np.array_equal(out[key], ref[key]) for key in ("x", "w1", "w2", "q1", "q2", "q3"))


def test_an_empty_run_async_leaves_the_run_events_alone(env):
    """Advisor, round 5: run_async(p, 0) enqueues nothing -- the two completion events keep standing for the runs they stand for, so
    runs_in_flight does not forget a run that is still running; and the blocking run() between two asynchronous ones does not re-arm
    an event either (only run_async binds the launch's stop event)."""
    import time

    flame_amd, oracle = env
    g = synth.make_graph("640x480", seed=43)
    ref = synth.copy_graph(g)
    p = flame_amd.Params()
    with flame_amd.Regularizer(0) as reg:
        reg.upload_graph(g)
        reg.run(p, 10)                      # (the topology's first launch is behind us)
        assert reg.runs_in_flight() == 0
        reg.run_async(p, 3000)              # ~3 ms
        n1 = reg.runs_in_flight()
        reg.run_async(p, 0)
        reg.run_async(p, 0)
        n2 = reg.runs_in_flight()
        assert n1 == 1 and n2 == 1, (n1, n2)
        reg.run_async(p, 3000)
        assert reg.runs_in_flight() == 2
        t0 = time.perf_counter()
        while reg.runs_in_flight() != 0:
            assert time.perf_counter() - t0 < 10.0
            time.sleep(0.0005)
        reg.run(p, 7)                       # blocking, between asynchronous ones
        reg.run_async(p, 500)
        assert reg.runs_in_flight() in (0, 1)
        reg.sync()
        assert reg.runs_in_flight() == 0
        oracle.run(ref, 10 + 6000 + 7 + 500)
        out = reg.download_state()
        assert all(np.array_equal(out[key], ref[key]) for key in ("x", "w1", "w2", "q1", "q2", "q3"))


def test_page_ranking_at_create_only_for_the_first_context_of_a_device(env):
    """Advisor, round 5: flame_nltgv2_create measures the record-placement pages (3 ms of cross-XCD spin kernels) only for the first
    context of a device, or takes a ranked pool that a closed context left behind; a further context created beside a live one ranks
    at its first run that places records -- never inside create, where it would compete with another context's free-running solver."""
    flame_amd, oracle = env
    g = synth.make_graph("640x480", seed=45)
    ref, _ = cpu_run(oracle, g, 40)
    a = flame_amd.Regularizer(0)
    try:
        # (earlier tests of this process have left ranked pools: `a` took one over, or measured -- either way it stands)
        assert a.placement_info()["state"] == 1
        pools_left = 0
        extra = []
        # drain the pools other tests left, so that the next create finds none
        for _ in range(64):
            r = flame_amd.Regularizer(0)
            extra.append(r)
            if r.placement_info()["state"] == 0:
                break
            pools_left += 1
        b = extra[-1]
        assert b.placement_info()["state"] == 0, "a context created beside live ones with no ranked pool free ranked its pages in create"
        b.upload_graph(g)
        b.run(flame_amd.Params(), 40)      # the first run that places records ranks them
        assert b.placement_info()["state"] == 1 and b.placement_info()["placed_records"] > 0
        assert_state_equal(b.download_state(), ref, what="lazily ranked context")
        for r in extra:
            r.close()
        c = flame_amd.Regularizer(0)        # ... and a closed context's pool, with its ranking, goes to the next one at create
        try:
            assert c.placement_info()["state"] == 1
        finally:
            c.close()
    finally:
        a.close()


@pytest.mark.parametrize("form", [3, 4, 6])
def test_export_target_switched_inside_a_replayed_chain(env, form):
    """Double-buffered gather rows: run k exports into row A, the target moves to row B, run k + 1 chains on.  If the chain
    is replayed (an expired wait), every run must be redone with the target IT was enqueued with -- row A gets run k's
    x * scale, not the previous frame's (advisor, round 3: PendingOp now records the target)."""
    import torch

    flame_amd, oracle = env
    g = synth.make_graph("320x240", seed=29)
    ref = synth.copy_graph(g)
    p = flame_amd.Params()
    rows = torch.full((2, g["V"]), -3.0, dtype=torch.float32, device="cuda")
    with flame_amd.Regularizer(0) as reg:
        reg.set_option(5, form)
        reg.upload_graph(g)
        reg.run(p, 10)
        oracle.run(ref, 10)
        reg.set_option(flame_amd.regularizer.OPT_FAULT_INJECT, 200)
        before = reg.info()["timeouts_recovered"]
        reg.set_export_target(rows[0].data_ptr(), 2.0)
        reg.run_async(p, 9)
        reg.set_export_target(rows[1].data_ptr(), 0.5)
        reg.run_async(p, 7)
        reg.sync()
        oracle.run(ref, 9)
        want_a = ref["x"] * np.float32(2.0)
        oracle.run(ref, 7)
        want_b = ref["x"] * np.float32(0.5)
        torch.cuda.synchronize()
        assert reg.info()["timeouts_recovered"] == before + 1
        got = rows.cpu().numpy()
        assert np.array_equal(got[0], want_a), "row A must hold the FIRST run's result after the replay"
        assert np.array_equal(got[1], want_b)
        assert_state_equal(reg.download_state(), ref, keys=OUT_KEYS, what="after the replayed chain")
        # the target in force after the replay is the caller's latest one
        reg.set_option(flame_amd.regularizer.OPT_FAULT_INJECT, 0)
        reg.run(p, 6)
        oracle.run(ref, 6)
        torch.cuda.synchronize()
        assert np.array_equal(rows[1].cpu().numpy(), ref["x"] * np.float32(0.5))
        assert np.array_equal(rows[0].cpu().numpy(), want_a)


def test_costs_summed_on_the_device_in_a_fixed_order(env):
    """FLAME_NLTGV2_OPT_COST_SUM = 1: smoothnessCost / dataCost (cc:51-85) with both sums formed on the device by k_block_sum -- 1024
    strided partial sums combined pairwise.  Bit-equal to the CPU restatement of that order over the reference's addends, and within
    1e-5 of the sequential sums of the default path (which stay bit-equal to the checker's)."""
    flame_amd, oracle = env
    for config in ("320x240", "1920x1080"):
        g = synth.make_graph(config, seed=4)
        p = flame_amd.Params()
        with flame_amd.Regularizer(0) as reg:
            reg.upload_graph(g)
            reg.run(p, 40)
            exact = reg.costs(p)
            reg.set_option(flame_amd.regularizer.OPT_COST_SUM, 1)
            fast = reg.costs(p)
            st = reg.download_state()
        ref = synth.copy_graph(g)
        oracle.run(ref, 40)
        assert exact == oracle.costs(ref)
        f = np.float32
        i, j = g["src"], g["dst"]
        dx, dy = g["pos"][i, 0] - g["pos"][j, 0], g["pos"][i, 1] - g["pos"][j, 1]
        a = np.abs(((st["x"][i] - st["x"][j]) - st["w1"][i] * dx) - st["w2"][i] * dy)
        terms = np.empty(2 * g["E"], f)
        terms[0::2] = g["alpha"] * a
        terms[1::2] = g["beta"] * np.abs(st["w1"][i] - st["w1"][j]) + g["beta"] * np.abs(st["w2"][i] - st["w2"][j])
        dterms = np.abs((st["x"] - g["data_term"]) * g["data_weight"]).astype(f)
        assert f(fast[0]) == f(p.data_factor) * f(oracle.strided_tree_sum(terms)), config
        assert f(fast[1]) == f(oracle.strided_tree_sum(dterms)), config
        assert abs(fast[0] - exact[0]) <= 1e-5 * abs(exact[0]) and abs(fast[1] - exact[1]) <= 1e-5 * abs(exact[1])


def test_run_timed_and_info(env):
    flame_amd, _ = env
    g = synth.make_graph("640x480", seed=1)
    with flame_amd.Regularizer(0) as reg:
        reg.upload_graph(g)
        ms = reg.run_timed(flame_amd.Params(), 200)
        info = reg.info()
    assert ms > 0
    assert info["V"] == g["V"] and info["E"] == g["E"]
    assert info["algorithmic_bytes_per_iter"] == 64 * g["V"] + 40 * g["E"]
    assert "gfx950" in info["gcn_arch"]


@pytest.mark.parametrize("trial", range(8))
def test_randomized_run_sequences(env, trial):
    """Random graphs, parameters, run lengths and option changes between runs (persistent forms, per-step path,
    canonical sweeps, state edits in between): the device state must track the checker bit for bit throughout."""
    import torch

    flame_amd, oracle = env
    rng = np.random.default_rng(400 + trial)
    if trial % 2:
        g = synth.make_graph(["320x240", "640x480"][trial % 4 // 2], seed=900 + trial)
    else:
        g = random_graph(int(rng.integers(50, 3000)), int(rng.integers(100, 9000)), seed=900 + trial)
    ref = synth.copy_graph(g)
    kw = dict(data_factor=float(rng.uniform(0.02, 0.5)), step_x=float(rng.uniform(2e-4, 5e-3)), step_q=float(rng.uniform(20, 300)),
              theta=float(rng.uniform(0.0, 1.0)), x_min=0.0, x_max=float(rng.uniform(2.0, 10.0)))
    p, rp = flame_amd.Params(**kw), oracle.make_params(**kw)
    buf = torch.zeros(g["V"], dtype=torch.float32, device="cuda")
    with flame_amd.Regularizer(0) as reg:
        reg.upload_graph(g)
        reg.set_export_target(buf.data_ptr(), 2.0)
        for step in range(12):
            form = int(rng.choice([0, 1, 3, 4, 6]))
            reg.set_option(5, form)
            reg.set_option(1, int(rng.random() < 0.15))      # canonical four-sweep path now and then
            reg.set_option(flame_amd.regularizer.OPT_DUAL_PUBLISH, int(rng.choice([0, 1, 2])))
            n = int(rng.choice([1, 2, 3, 5, 8, 13, 40, 120]))
            reg.run(p, n)
            assert oracle.run(ref, n, rp) == 0
            if rng.random() < 0.3:                            # the caller edits the data term between runs
                ref["data_term"] = (ref["data_term"] * np.float32(rng.uniform(0.95, 1.05))).astype(np.float32)
                ref["data_weight"] = (0.5 + rng.random(g["V"])).astype(np.float32)
                reg.update_data(ref["data_term"], ref["data_weight"])
            if step % 4 == 3:
                assert_state_equal(reg.download_state(), ref, keys=OUT_KEYS + ("x_prev", "w1_prev", "w2_prev"), what=f"trial {trial} step {step}")
                torch.cuda.synchronize()
        assert_state_equal(reg.download_state(), ref, keys=OUT_KEYS + ("x_prev", "w1_prev", "w2_prev"), what=f"trial {trial} end")
        torch.cuda.synchronize()
        assert np.array_equal(buf.cpu().numpy(), ref["x"] * np.float32(2.0))  # the standing export target followed every path
        sm, dc = reg.costs(p)
        rs, rd = oracle.costs(ref, rp)
        assert np.float32(sm) == np.float32(rs) and np.float32(dc) == np.float32(rd)


@pytest.mark.gpu
@pytest.mark.parametrize("config", ["320x240", "640x480", "1280x720"])
def test_record_placement_is_bit_identical_and_well_formed(env, config):
    """FLAME_NLTGV2_OPT_PLACEMENT (on by default): the records another XCD reads live in a pool of pages ranked per pair of
    XCDs.  Same bits with and without, and with the run-time record verification on (whose second read goes through the
    placed address too); the placed offsets are aligned, inside the pool, unique, and exactly the records the host's own
    patch walk says cross an XCD border (flame_nltgv2_layout_selftest)."""
    flame_amd, oracle = env
    from flame_amd.regularizer import OPT_PERSISTENT, OPT_PLACEMENT, OPT_VERIFY_RECORDS
    g = synth.make_graph(config, seed=11)
    ref, _ = cpu_run(oracle, g, 60)
    for place, verify in ((1, 0), (0, 0), (1, 1)):
        with flame_amd.Regularizer(0) as reg:
            reg.set_option(OPT_PERSISTENT, 4)
            reg.set_option(OPT_PLACEMENT, place)
            reg.set_option(OPT_VERIFY_RECORDS, verify)
            reg.upload_graph(g)
            reg.run(flame_amd.Params(), 30)
            reg.run(flame_amd.Params(), 30)
            assert_state_equal(reg.download_state(), ref, what=f"placement {place} verify {verify}")
            info, pi = reg.info(), reg.placement_info()
            assert info["last_run_path"] == 6 and info["timeouts_recovered"] == 0 and info["torn_records_detected"] == 0
            if place:
                assert pi["state"] == 1 and pi["placed_records"] > 0, pi
                assert 0.0 < pi["best_us"] <= pi["mean_us"] <= pi["worst_us"] < 5.0, pi
                assert reg.layout_selftest() == 0
            else:
                assert pi["placed_records"] == 0, pi  # (the page ranking itself is measured when a context is created: state 1 either way)


@pytest.mark.gpu
def test_after_an_expired_run_the_persistent_path_is_tried_again(env):
    """An expired run does not leave a static graph on the per-step path for good: the next 4 runs of the topology go per step, then
    the persistent launch is tried again (8 after a second expired run in a row, ... 1024 at most)."""
    flame_amd, oracle = env
    from flame_amd.regularizer import OPT_FAULT_INJECT, OPT_VERIFY_RECORDS

    g = synth.make_graph("320x240", seed=77)
    p = flame_amd.Params()
    ref = synth.copy_graph(g)
    with flame_amd.Regularizer(0) as reg:
        reg.upload_graph(g)
        reg.set_option(OPT_VERIFY_RECORDS, 2)   # test hook: one corrupted re-read in the next persistent run -> taken back like an expired run
        reg.run(p, 12)
        oracle.run(ref, 12)
        assert reg.info()["torn_records_detected"] == 1 and reg.info()["last_run_path"] in (2, 3)
        paths = []
        for _ in range(6):
            reg.run(p, 12)
            oracle.run(ref, 12)
            paths.append(reg.info()["last_run_path"])
        # four runs per step, then the persistent launch again -- where the hook fires once more (8 runs of back-off now)
        assert paths[:4] == [paths[0]] * 4 and paths[0] in (2, 3), paths
        assert reg.info()["torn_records_detected"] == 2, (paths, reg.info()["torn_records_detected"])
        reg.set_option(OPT_VERIFY_RECORDS, 0)
        reg.run(p, 12)
        oracle.run(ref, 12)
        assert reg.info()["last_run_path"] == 6
        assert_state_equal(reg.download_state(), ref, keys=OUT_KEYS, what="after the back-off")


@pytest.mark.gpu
def test_an_expired_run_at_high_residency_makes_the_planner_leave_room(env):
    """Ten frames of 640x480 take 18.6 of the 20 wave slots a CU really holds for the two-half-edges-per-lane form.  When such a run
    expires (here: fault injection; in a pipeline: other kernels keeping slots busy), the following topologies are planned for at
    most 16 waves per CU -- the same form in two groups of five frames for this graph -- instead of trying the same launch frame after frame; a
    1080p frame (12.6 waves per CU) and a small graph are not affected.  Everything stays bit-identical to the checker."""
    flame_amd, oracle = env
    from flame_amd.regularizer import OPT_FAULT_INJECT

    g = synth.concat_graphs([synth.make_graph("640x480", seed=12 + i) for i in range(10)])
    p = flame_amd.Params()
    ref = synth.copy_graph(g)
    with flame_amd.Regularizer(0) as reg:
        reg.upload_graph(g)
        reg.run(p, 20)
        oracle.run(ref, 20)
        assert reg.info()["last_run_path"] == 7 and reg.info()["last_run_groups"] == 1
        reg.set_option(OPT_FAULT_INJECT, 200)
        reg.run(p, 21)                       # expires, taken back, redone with room left on every CU: two groups of five frames
        oracle.run(ref, 21)
        assert reg.info()["timeouts_recovered"] == 1 and reg.info()["last_run_path"] == 7 and reg.info()["last_run_groups"] == 2, reg.info()
        reg.set_option(OPT_FAULT_INJECT, 0)
        for k in range(2):                   # new topologies (the same graph uploaded again): planned with room to spare
            st = reg.download_state()
            g2 = synth.copy_graph(g)
            g2.update({key: st[key] for key in st})
            reg.upload_graph(g2)
            reg.run(p, 10 + k)
            oracle.run(ref, 10 + k)
            assert reg.info()["last_run_path"] == 7 and reg.info()["last_run_groups"] == 2, reg.info()
        assert_state_equal(reg.download_state(), ref, keys=OUT_KEYS, what="after the crowded topologies")
        assert reg.info()["timeouts_recovered"] == 1
    big = synth.make_graph("1920x1080", seed=12)
    with flame_amd.Regularizer(0) as reg:    # 12.6 waves per CU: an expired run changes nothing for the next topology
        reg.upload_graph(big)
        reg.set_option(OPT_FAULT_INJECT, 200)
        reg.run(p, 20)
        assert reg.info()["timeouts_recovered"] == 1
        reg.set_option(OPT_FAULT_INJECT, 0)
        reg.upload_graph(big)
        reg.run(p, 20)
        assert reg.info()["last_run_path"] == 7
    small = synth.make_graph("640x480", seed=12)
    with flame_amd.Regularizer(0) as reg:    # 4 patches per CU: an expired run changes nothing for the next topology
        reg.upload_graph(small)
        reg.set_option(OPT_FAULT_INJECT, 200)
        reg.run(p, 20)
        reg.set_option(OPT_FAULT_INJECT, 0)
        reg.upload_graph(small)
        reg.run(p, 20)
        assert reg.info()["last_run_path"] == 6


@pytest.mark.parametrize("config,form", [("640x480", 1), ("1920x1080", 1), ("320x240", 4)])
def test_an_expired_chain_is_first_redone_persistently_then_per_step(env, config, form):
    """The rung between the persistent forms and the one-launch-per-step path (5x slower): a chain whose wait expired is taken back and
    redone in a persistent form planned for at most 16 waves per CU (a 1080p frame: the two-half-edges form); only if THAT expires as
    well (fault 2^22 + n: the hook hits the replay too) do the steps go one launch at a time.  Bit-identical either way."""
    from flame_amd.regularizer import OPT_FAULT_INJECT, OPT_PERSISTENT

    flame_amd, oracle = env
    g = synth.make_graph(config, seed=31)
    ref = synth.copy_graph(g)
    p = flame_amd.Params()
    with flame_amd.Regularizer(0) as reg:
        reg.set_option(OPT_PERSISTENT, form)
        reg.upload_graph(g)
        reg.run(p, 20)
        oracle.run(ref, 20)
        clean_path = reg.info()["last_run_path"]
        assert clean_path in (6, 7)
        reg.set_option(OPT_FAULT_INJECT, 300)
        reg.run_async(p, 30)
        reg.run_async(p, 11)
        reg.sync()                             # the chain expires in its first run, is redone persistently
        oracle.run(ref, 41)
        info = reg.info()
        assert info["timeouts_recovered"] == 1 and info["last_run_path"] in (6, 7), info
        assert_state_equal(reg.download_state(), ref, keys=OUT_KEYS + ("x_prev", "w1_prev", "w2_prev"), what="after the persistent replay")
        reg.set_option(OPT_FAULT_INJECT, 0)    # (lets the next run be persistent again: the back-off is lifted with the hook)
        reg.set_option(OPT_FAULT_INJECT, (1 << 22) + 300)
        reg.run_async(p, 25)
        reg.run_async(p, 7)
        reg.sync()                             # expires, its persistent replay expires too, redone one launch per step
        oracle.run(ref, 32)
        info = reg.info()
        assert info["timeouts_recovered"] == 2 and info["last_run_path"] in (2, 3), info
        assert_state_equal(reg.download_state(), ref, keys=OUT_KEYS + ("x_prev", "w1_prev", "w2_prev"), what="after the per-step replay")


@pytest.mark.parametrize("size,path", [("640x480", 6), ("1920x1080", 7)])
def test_open_run_stops_when_asked_and_says_how_far_it_went(built, size, path):
    """flame_nltgv2_run_open: ONE launch that iterates until the next call that needs the solver settled asks it to stop -- a patch reads the
    request and publishes the iteration every patch leaves at.  However far it went (flame_nltgv2_iterations says), the state is the
    oracle's after exactly that many iterations: stopped after a few milliseconds, run to its bound, stopped at once, settled by a run
    enqueued behind it, not applicable (record verification on: nothing enqueued), and expired (the test hook's fault: taken back, redone)."""
    import time

    import torch  # noqa: F401

    import flame_amd
    from flame_amd.regularizer import OPT_FAULT_INJECT, OPT_VERIFY_RECORDS
    from oracle import capi as oracle

    keys = OUT_KEYS + ("x_prev", "w1_prev", "w2_prev")
    g = synth.make_graph(size, seed=3)  # (640x480: the patch-per-wave form; 1920x1080: two half-edges per lane)
    ref = synth.copy_graph(g)
    p = flame_amd.Params()
    with flame_amd.Regularizer(0) as reg:
        reg.upload_graph(g)
        reg.run(p, 10)
        oracle.run(ref, 10)
        assert reg.iterations() == (10, False)

        def settle_and_check(what, settle, lo, hi, extra=0):
            before = reg.iterations()[0]
            assert reg.iterations()[1]
            out = settle()
            total, still_open = reg.iterations()
            n = total - before - extra
            assert not still_open and n % 2 == 0 and lo <= n <= hi, (what, n)
            oracle.run(ref, n + extra)
            assert_state_equal(out if out is not None else reg.download_state(), ref, keys=keys, what=f"{what}: {n} iterations")
            return n

        assert reg.run_open(p, 400000)
        assert reg.runs_in_flight() >= 1
        time.sleep(0.004)
        n = settle_and_check("stopped after 4 ms by a read-back", lambda: reg.download_state(), 500, 40000)
        assert reg.info()["last_run_path"] == path
        assert reg.run_open(p, 64)
        time.sleep(0.002)
        settle_and_check("run to its bound", lambda: reg.sync(), 64, 64)
        assert reg.run_open(p, 400000)
        n0 = settle_and_check("stopped at once", lambda: reg.sync(), 2, 4000)
        assert reg.run_open(p, 400000)
        time.sleep(0.001)
        settle_and_check("settled by a run enqueued behind it", lambda: (reg.run_async(p, 30), reg.sync())[1], 2, 40000, extra=30)
        # odd bounds are refused; with the record verification on an open run is not applicable: nothing happens
        with pytest.raises(flame_amd.NLTGV2Error):
            reg.run_open(p, 33)
        reg.set_option(OPT_VERIFY_RECORDS, 1)
        before = reg.iterations()
        assert not reg.run_open(p, 1000) and reg.iterations() == before
        reg.set_option(OPT_VERIFY_RECORDS, 0)
        # an open run whose waits expire: taken back and redone
        rec = reg.info()["timeouts_recovered"]
        reg.set_option(OPT_FAULT_INJECT, 200)
        assert reg.run_open(p, 400000)
        time.sleep(0.002)
        settle_and_check("expired, redone", lambda: reg.sync(), 2, 40000)
        reg.set_option(OPT_FAULT_INJECT, 0)
        assert reg.info()["timeouts_recovered"] == rec + 1
        print(f"open runs: {n} iterations in ~4 ms, {n0} when stopped at once")
