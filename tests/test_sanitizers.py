"""The library's HOST code under AddressSanitizer + UndefinedBehaviorSanitizer (`make -C flame_amd/csrc sanitize`): the
Delaunay triangulator (flame_amd/csrc/delaunay.cpp), the layout builders and the CPU replay of the patch rows
(flame_amd/csrc/nltgv2_pack.hpp via tests/cpp/wg_layout_test.cc and tests/cpp/host_sanitize_test.cc) and the facade's packing
(include/flame_hip/*.hpp).  No device needed; any out-of-bounds access, use after free, signed overflow or invalid shift
aborts the programs (-fno-sanitize-recover=all)."""
import os
import subprocess

from tests.conftest import ROOT


def test_host_code_is_clean_under_asan_and_ubsan(built):
    r = subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "flame_amd", "csrc"), "sanitize"], capture_output=True, text=True,
                       timeout=900)
    out = r.stdout + r.stderr
    assert r.returncode == 0, out[-4000:]
    assert "sanitized host code: all ok" in out and "all ok" in out.split("sanitized host code: all ok")[1], out[-4000:]
    assert "runtime error" not in out and "AddressSanitizer" not in out, out[-4000:]
