"""The C++ facade (include/flame_hip/*.hpp: the reference's namespace/function/struct names over the
C-ABI).  CPU part: it compiles as plain C++11 with g++ and links against the in-tree library.  GPU
part: the built program runs step()/run()/costs through the facade and compares with the checker."""
import os
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
