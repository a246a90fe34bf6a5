"""The C++ facade (include/flame_hip/*.hpp: the reference's namespace/function/struct names over the
C-ABI).  CPU part: it compiles as plain C++11 with g++ and links against the in-tree library.  GPU
part: the built program runs step()/run()/costs through the facade and compares with the checker."""
import os
import re
import subprocess

import pytest

from tests.conftest import ROOT


def build_program(tmp_path):
    exe = str(tmp_path / "facade_test")
    lib_dir = os.path.join(ROOT, "flame_amd")
    subprocess.check_call([
        "g++", "-std=c++11", "-O1", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(ROOT, "include"),
        os.path.join(ROOT, "tests", "cpp", "facade_test.cc"), "-o", exe,
        "-L", lib_dir, "-lflame_nltgv2_hip", "-L", os.path.join(ROOT, "oracle"), "-loracle_nltgv2",
        f"-Wl,-rpath,{lib_dir}", f"-Wl,-rpath,{os.path.join(ROOT, 'oracle')}", "-Wl,-rpath,/opt/rocm/lib"])
    return exe


def build_bgl_program(tmp_path):
    exe = str(tmp_path / "bgl_adaptor_test")
    lib_dir = os.path.join(ROOT, "flame_amd")
    subprocess.check_call([
        "g++", "-std=c++11", "-O1", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(ROOT, "tests", "cpp", "mock_boost"),
        "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cpp", "bgl_adaptor_test.cc"), "-o", exe,
        "-L", lib_dir, "-lflame_nltgv2_hip", "-L", os.path.join(ROOT, "oracle"), "-loracle_nltgv2",
        f"-Wl,-rpath,{lib_dir}", f"-Wl,-rpath,{os.path.join(ROOT, 'oracle')}", "-Wl,-rpath,/opt/rocm/lib"])
    return exe


def build_tracker_program(tmp_path):
    exe = str(tmp_path / "feature_tracker_test")
    lib_dir = os.path.join(ROOT, "flame_amd")
    subprocess.check_call([
        "g++", "-std=c++11", "-O1", "-Wall", "-Wextra", "-Werror", "-Wno-invalid-offsetof", "-I", os.path.join(ROOT, "include"),
        os.path.join(ROOT, "tests", "cpp", "feature_tracker_test.cc"), "-o", exe,
        "-L", lib_dir, "-lflame_nltgv2_hip", "-L", os.path.join(ROOT, "oracle"), "-loracle_nltgv2",
        f"-Wl,-rpath,{lib_dir}", f"-Wl,-rpath,{os.path.join(ROOT, 'oracle')}", "-Wl,-rpath,/opt/rocm/lib"])
    return exe


def test_feature_tracker_binding_compiles_and_fails_loudly_without_a_device(built, tmp_path):
    """include/flame_hip/feature_tracker.hpp with look-alikes of the reference's Params/Frame/SE3/FeatureWithIDepth."""
    exe = build_tracker_program(tmp_path)
    from tests.conftest import HAS_GPU

    if not HAS_GPU:
        r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
        assert r.returncode == 77 and "no usable HIP device" in r.stdout, r.stdout + r.stderr


@pytest.mark.gpu
def test_feature_tracker_binding_end_to_end(built, tmp_path):
    r = subprocess.run([build_tracker_program(tmp_path)], capture_output=True, text=True, timeout=300)
    print(r.stdout, r.stderr)
    assert r.returncode == 0 and r.stdout.count(" ok") >= 3, r.stdout + r.stderr


def test_bgl_adaptor_compiles_against_the_bgl_api(built, tmp_path):
    """include/flame_hip/bgl_adaptor.hpp against a test-only mock of the Boost.Graph calls it makes."""
    assert os.path.exists(build_bgl_program(tmp_path))


@pytest.mark.gpu
def test_bgl_adaptor_end_to_end(built, tmp_path):
    r = subprocess.run([build_bgl_program(tmp_path)], capture_output=True, text=True, timeout=300)
    print(r.stdout, r.stderr)
    assert r.returncode == 0 and "ok" in r.stdout, r.stdout + r.stderr


def test_facade_compiles_as_cxx11_and_links(built, tmp_path):
    assert os.path.exists(build_program(tmp_path))


@pytest.mark.gpu
def test_facade_end_to_end(built, tmp_path):
    exe = build_program(tmp_path)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    print(r.stdout, r.stderr)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.count(" ok") >= 5


def build_solver_loop_program(tmp_path):
    exe = str(tmp_path / "solver_loop_test")
    lib_dir = os.path.join(ROOT, "flame_amd")
    subprocess.check_call([
        "g++", "-std=c++11", "-O1", "-Wall", "-Wextra", "-Werror", "-pthread", "-I", os.path.join(ROOT, "include"),
        os.path.join(ROOT, "tests", "cpp", "solver_loop_test.cc"), "-o", exe,
        "-L", lib_dir, "-lflame_nltgv2_hip", "-L", os.path.join(ROOT, "oracle"), "-loracle_nltgv2",
        f"-Wl,-rpath,{lib_dir}", f"-Wl,-rpath,{os.path.join(ROOT, 'oracle')}", "-Wl,-rpath,/opt/rocm/lib"])
    return exe


def test_solver_loop_compiles_and_fails_loudly_without_a_device(built, tmp_path):
    """include/flame_hip/solver_loop.hpp: the solver thread of flame.cc:99-112 with a stop flag (C++11, -pthread)."""
    exe = build_solver_loop_program(tmp_path)
    from tests.conftest import HAS_GPU

    if not HAS_GPU:
        r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
        assert r.returncode == 77 and "no usable HIP device" in r.stdout, r.stdout + r.stderr


@pytest.mark.gpu
def test_solver_loop_end_to_end(built, tmp_path):
    """start, three frames of edit -> dirty -> resume (bit-identical to the checker), the stale-graph guard, stop and join."""
    r = subprocess.run([build_solver_loop_program(tmp_path)], capture_output=True, text=True, timeout=300)
    print(r.stdout, r.stderr)
    assert r.returncode == 0 and r.stdout.count(" ok") >= 8 and "FAIL" not in r.stdout, r.stdout + r.stderr


def build_integration_switch_program(tmp_path):
    """every fenced block behind an `<!-- edit:NAME ... -->` tag, written out verbatim as edit_NAME.inc and
    #included by tests/cpp/integration_switch_test.cc at the place of a mock Flame (members typed as flame.h:512, 536-539) it names."""
    import re

    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    blocks = dict(re.findall(r"<!-- edit:(\w+)[^>]*-->\n```cpp\n(.*?)```", text, flags=re.S))
    assert sorted(blocks) == ["1a", "1b", "2", "3", "4"], sorted(blocks)
    for name, body in blocks.items():
        (tmp_path / f"edit_{name}.inc").write_text(body)
    exe = str(tmp_path / "integration_switch_test")
    lib_dir = os.path.join(ROOT, "flame_amd")
    subprocess.check_call([
        "g++", "-std=c++11", "-O1", "-Wall", "-Wextra", "-Werror", "-pthread", "-I", str(tmp_path),
        "-I", os.path.join(ROOT, "tests", "cpp", "mock_boost"), "-I", os.path.join(ROOT, "include"),
        os.path.join(ROOT, "tests", "cpp", "integration_switch_test.cc"), "-o", exe, "-L", lib_dir, "-lflame_nltgv2_hip",
        f"-Wl,-rpath,{lib_dir}", "-Wl,-rpath,/opt/rocm/lib"])
    return exe


def test_integration_switch_compiles_verbatim(built, tmp_path):
    """The documented switch compiles as written against the reference's member types: `std::mutex graph_mtx_` (flame.h:539),
    `std::recursive_mutex update_mtx_` (flame.h:512), `Graph graph_` (flame.h:536); SolverLoop<Graph>'s default mutex is std::mutex."""
    exe = build_integration_switch_program(tmp_path)
    from tests.conftest import HAS_GPU

    if not HAS_GPU:
        r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
        assert r.returncode == 77 and "no usable HIP device" in r.stdout, r.stdout + r.stderr


@pytest.mark.gpu
def test_integration_switch_end_to_end(built, tmp_path):
    """Four update() calls of the mock Flame against the free-running solver, then ~Flame joining it with graph_mtx_ held."""
    r = subprocess.run([build_integration_switch_program(tmp_path)], capture_output=True, text=True, timeout=300)
    print(r.stdout, r.stderr)
    assert r.returncode == 0 and "FAIL" not in r.stdout and "~Flame with graph_mtx_ (std::mutex) held: ok" in r.stdout, r.stdout + r.stderr


def build_gather_program(tmp_path):
    exe = str(tmp_path / "frame_gather_test")
    lib_dir = os.path.join(ROOT, "flame_amd")
    subprocess.check_call([
        "g++", "-std=c++11", "-O1", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(ROOT, "include"),
        os.path.join(ROOT, "tests", "cpp", "frame_gather_test.cc"), "-o", exe,
        "-L", lib_dir, "-lflame_nltgv2_hip", "-L", os.path.join(ROOT, "oracle"), "-loracle_nltgv2", "-L/opt/rocm/lib", "-lamdhip64",
        f"-Wl,-rpath,{lib_dir}", f"-Wl,-rpath,{os.path.join(ROOT, 'oracle')}", "-Wl,-rpath,/opt/rocm/lib"])
    return exe


def test_frame_gather_compiles_and_fails_loudly_without_a_device(built, tmp_path):
    """include/flame_hip/frame_gather.hpp + include/flame_frames.h: the C++ host's multi-GPU gather (RCCL, bound at run time)."""
    exe = build_gather_program(tmp_path)
    from tests.conftest import HAS_GPU

    if not HAS_GPU:
        r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
        assert r.returncode == 77, r.stdout + r.stderr


@pytest.mark.gpu
def test_frame_gather_end_to_end(built, tmp_path):
    """Configuration 4 from C++: one solver per visible GPU, device-side export, one grouped ncclAllGather, compared
    with the checker on every device."""
    r = subprocess.run([build_gather_program(tmp_path)], capture_output=True, text=True, timeout=300,
                       env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
    print(r.stdout, r.stderr)
    assert r.returncode == 0 and "RCCL all-gather over" in r.stdout and "FAIL" not in r.stdout, r.stdout + r.stderr


# ---- the frame loop in C++ (Flame::update()'s order, flame.cc:265-415) with the solver free-running beside it ------------------------
def build_frame_loop_program(tmp_path):
    exe = str(tmp_path / "frame_loop_test")
    lib_dir = os.path.join(ROOT, "flame_amd")
    subprocess.check_call([
        "g++", "-std=c++11", "-O1", "-ffp-contract=off", "-pthread", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(ROOT, "include"),
        os.path.join(ROOT, "tests", "cpp", "frame_loop_test.cc"), "-o", exe, "-L", lib_dir, "-lflame_nltgv2_hip",
        f"-Wl,-rpath,{lib_dir}", "-Wl,-rpath,/opt/rocm/lib"])
    return exe


def test_frame_loop_compiles(built, tmp_path):
    """tests/cpp/frame_loop_test.cc against the facade (SolverLoop's device mode, syncPrepare / syncCommit, interpolateMeshBegin / End,
    FeatureTracker, delaunayTriangulate): builds with -Wall -Wextra -Werror; without a GPU it says so and exits with 77."""
    exe = build_frame_loop_program(tmp_path)
    from tests.conftest import HAS_GPU

    if not HAS_GPU:
        r = subprocess.run([exe, "/dev/null", "/dev/null"], capture_output=True, text=True, timeout=120)
        assert r.returncode == 77 and "no usable HIP device" in r.stdout, r.stdout + r.stderr


@pytest.mark.gpu
def test_frame_loop_end_to_end(built, tmp_path):
    """Twelve frames of FeatureTracker::updateFeatureIDepths -> delaunayTriangulate -> projectGraph -> DeviceGraph::syncPrepare (the
    solver thread iterating on) -> syncCommit -> interpolateMeshBegin / End, driven from C++ with flame_hip::SolverLoop free-running in
    device mode.  The program logs how many iterations the loop had applied at every call that touches the device image; the chained
    CPU checkers replay exactly those iterations between the same edits: the features after every update, every keep mask, every dense
    map with its coverage and the graph's state at the end of every frame must be the checkers' bit for bit."""
    import struct

    import numpy as np

    import flame_amd
    from flame_amd import synth
    from flame_amd import synth_stereo as ss
    from oracle import capi as oracle
    from oracle import stereo_capi as so
    from tests import test_pipeline as tp
    from tests.helpers import OUT_KEYS

    n_new = 12
    new_frames = tuple(range(20, 20 + n_new))
    sc = ss.PlaneScene(tp.W, tp.H, seed=21, normal=(0.2, -0.1, 1.0), distance=2.2)
    sc.add_camera(10, np.eye(3), [0, 0, 0])
    sc.add_camera(11, ss.rot([0, 1, 0], 0.004), [-0.03, 0.002, -0.005])
    for i, k in enumerate(new_frames):
        sc.add_camera(k, ss.rot([0.1, 1, 0.05], 0.008 + 0.002 * i), [-0.07 - 0.015 * i, 0.004 + 0.001 * i, -0.015 - 0.004 * i])
    imgs = {c: sc.render(c) for c in sc.cams}
    feats = ss.make_features(sc, so.FEATURE_DTYPE, [10, 11], 500, 21, mu_noise=0.12, var=0.03)
    side = tp.OracleSide(sc, imgs)
    side.add_frame(10), side.add_frame(11)
    # ---- pass 1, checkers only: the features after every frame and the feature set that enters the graph -------------------------
    blob = [struct.pack("<7i", tp.W, tp.H, tp.PAD, len(feats), 2, n_new, 300), sc.K32.astype("<f4").tobytes(), sc.Kinv32.astype("<f4").tobytes()]
    for c in (10, 11):
        blob += [struct.pack("<I", c), np.ascontiguousarray(imgs[c], np.uint8).tobytes()]
    blob.append(feats.tobytes())
    want_feats, want_stats, graph_in, prev = [], [], [], None
    f = feats.copy()
    for k in new_frames:
        side.add_frame(k)
        want_stats.append(side.update_features(f, k))
        want_feats.append(f.tobytes())
        fid, pos, idepth = tp.project_features(sc, f, k)
        assert len(fid) > 150
        blob += [struct.pack("<II", k, 11), np.ascontiguousarray(imgs[k], np.uint8).tobytes()]
        poses = ss.poses_for(sc, [10, 11], k, 11)
        blob.append(struct.pack("<i", len(poses)))
        for p in poses:
            blob.append(struct.pack("<I", p["id"]) + np.concatenate([p["q_to_new"], p["t_to_new"], p["q_to_pf"], p["t_to_pf"]]).astype("<f4").tobytes())
        blob += [struct.pack("<i", len(fid)), fid.astype("<i4").tobytes(), pos.astype("<f4").tobytes(), idepth.astype("<f4").tobytes()]
        proj = None
        if prev is None:
            blob.append(struct.pack("<i", 0))
        else:
            q, t, KRKinv = tp.projection_between(sc, prev, k)
            proj = (q, t, KRKinv)
            blob.append(struct.pack("<i", 1) + np.concatenate([sc.K32.ravel(), sc.Kinv32.ravel(), KRKinv.ravel(), np.asarray(q, np.float32),
                                                                np.asarray(t, np.float32), np.asarray(tp.REGION, np.float32)]).astype("<f4").tobytes())
        graph_in.append((fid, pos, idepth, proj))
        prev = k
    fin, fout = str(tmp_path / "frames.bin"), str(tmp_path / "log.bin")
    with open(fin, "wb") as fh:
        fh.write(b"".join(blob))
    exe = build_frame_loop_program(tmp_path)
    keep = os.environ.get("FLAME_KEEP_FRAME_LOOP")  # (tools/r06_frame_loop_gaps.sh: the program and its input, for a profiler)
    if keep:
        import shutil

        os.makedirs(keep, exist_ok=True)
        shutil.copy(exe, os.path.join(keep, "frame_loop_test")), shutil.copy(fin, os.path.join(keep, "frames.bin"))
    r = subprocess.run([exe, fin, fout, "200"], capture_output=True, text=True, timeout=600)
    print(r.stdout, r.stderr)
    assert r.returncode == 0 and "FAIL" not in r.stdout and "solver busy" in r.stdout, r.stdout + r.stderr
    # (the same frames once more with the test's window closed -- no state read-back, the map copied after the device is released: the
    #  solver-idle share of the loop as FLaME would drive it; its log is not replayed)
    r2 = subprocess.run([exe, fin, str(tmp_path / "log_lean.bin"), "200", "1"], capture_output=True, text=True, timeout=600)
    print(r2.stdout, r2.stderr)
    assert r2.returncode == 0 and "FAIL" not in r2.stdout and "solver busy" in r2.stdout, r2.stdout + r2.stderr
    # (the mesh of a frame is begun in the second part of a two-part hold -- SolverLoop::withDevice(f, g), FLAME_NLTGV2_OPT_MESH_STATE = 1 --
    #  and must not have settled the rounds enqueued between the parts: counted by the program, at least half the frames of both runs)
    for out in (r.stdout, r2.stdout):
        m = re.search(r"meshes begun beside the next rounds: (\d+) of (\d+)", out)
        assert m and int(m.group(2)) >= 10 and 2 * int(m.group(1)) >= int(m.group(2)), out  # (a short round may have ended by itself)
    # ---- pass 2: the program's log replayed on the chained checkers ------------------------------------------------------------------
    log = open(fout, "rb").read()
    at = [0]

    def take(dtype, n):
        a = np.frombuffer(log, dtype=dtype, count=n, offset=at[0])
        at[0] += a.nbytes
        return a

    it_prev, total_iters = 0, 0
    for i, k in enumerate(new_frames):
        fid, pos, idepth, proj = graph_in[i]
        got_feats = take(np.uint8, len(feats) * so.FEATURE_DTYPE.itemsize).tobytes()
        got_stats = take("<i4", 8)
        assert got_feats == want_feats[i], f"frame {k}: features after updateFeatureIDepths"
        assert [int(v) for v in got_stats[:6]] == want_stats[i], (k, got_stats, want_stats[i])
        tris = take("<i4", 3 * int(take("<i4", 1)[0])).reshape(-1, 3).copy()
        edges = take("<i4", 2 * int(take("<i4", 1)[0])).reshape(-1, 2).copy()
        t2, e2 = flame_amd.delaunay(pos)
        assert np.array_equal(tris, t2) and np.array_equal(edges, e2), f"frame {k}: triangulation"
        if proj is None:
            assert int(take("<u8", 1)[0]) == 0
            side.first_graph(synth.assemble_graph(pos, idepth, edges), fid)
        else:
            it_project = int(take("<u8", 1)[0])
            keep = take(np.uint8, int(take("<i4", 1)[0])).copy()
            it_commit = int(take("<u8", 1)[0])
            side.run(it_project - it_prev)
            assert np.array_equal(side.project_graph(*proj), keep), f"frame {k}: projectGraph keep mask"
            side.run(it_commit - it_project)          # the solver went on between syncPrepare and syncCommit: on the projected graph
            side.sync_graph(fid, pos, idepth, edges)
            it_prev = it_commit
        it_raster = int(take("<u8", 1)[0])
        coverage = int(take("<i4", 1)[0])
        dense = take("<f4", tp.W * tp.H).reshape(tp.H, tp.W)
        side.run(it_raster - it_prev)
        want = side.interpolate(tris)
        assert np.array_equal(dense, want, equal_nan=True), f"frame {k}: dense inverse depth map"
        assert coverage == oracle.raster_coverage(np.ascontiguousarray(want)), (k, coverage)
        it_state = int(take("<u8", 1)[0])
        V, E = (int(v) for v in take("<i4", 2))
        state = {n: take("<f4", V).copy() for n in ("x", "w1", "w2", "x_bar", "w1_bar", "w2_bar")}
        state.update({n: take("<f4", E).copy() for n in ("q1", "q2", "q3")})
        side.run(it_state - it_raster)
        ref = side.state()
        assert ref["x"].shape[0] == V and ref["q1"].shape[0] == E
        for n in OUT_KEYS:
            assert np.array_equal(state[n], ref[n]), f"frame {k}: {n} after {it_state} iterations of the free-running solver"
        total_iters += it_state - it_prev if proj is None else 0
        it_prev = it_state
    assert at[0] == len(log)
    assert it_prev >= 200 * n_new, f"the solver iterated only {it_prev} times beside {n_new} frames"
    tp.check_against_truth.__globals__["NEW_FRAMES"] = new_frames  # (the truth check of the pipeline test, at THIS last frame)
    try:
        tp.check_against_truth(sc, dense)
    finally:
        tp.check_against_truth.__globals__["NEW_FRAMES"] = tp.NEW_FRAMES
