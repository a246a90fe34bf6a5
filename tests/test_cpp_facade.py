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


def test_facade_compiles_as_cxx11_and_links(built, tmp_path):
    assert os.path.exists(build_program(tmp_path))


@pytest.mark.gpu
def test_facade_end_to_end(built, tmp_path):
    exe = build_program(tmp_path)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    print(r.stdout, r.stderr)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.count(" ok") >= 5
