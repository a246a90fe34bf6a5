"""The C-ABI library loads here (no GPU) and exports every symbol include/*.h declares;
without a device every compute entry point fails loudly (no CPU fallback in the product path)."""
import ctypes as C
import os
import re
import subprocess

import pytest

from tests.conftest import HAS_GPU, ROOT

HEADERS = [os.path.join(ROOT, "include", h) for h in ("flame_nltgv2.h", "flame_stereo.h", "flame_frames.h")]


FRAMES_ABI_SYMBOLS = ("flame_frames_create", "flame_frames_destroy", "flame_frames_count", "flame_frames_local_row",
                      "flame_frames_stream", "flame_frames_gather", "flame_frames_wait", "flame_frames_gathered",
                      "flame_frames_download", "flame_frames_last_error_text")


def declared_symbols():
    out = set()
    for h in HEADERS:
        txt = open(h).read()
        txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
        out |= set(re.findall(r"\b(flame_(?:nltgv2|delaunay|stereo|frames)_[a-z_0-9]+)\s*\(", txt))
    return sorted(out)


def test_header_symbols_are_exported(built):
    import flame_amd
    from flame_amd.regularizer import ABI_SYMBOLS
    from flame_amd.stereo import STEREO_ABI_SYMBOLS

    STEREO_ABI_SYMBOLS = tuple(STEREO_ABI_SYMBOLS) + FRAMES_ABI_SYMBOLS

    decl = declared_symbols()
    assert len(decl) >= 38
    assert set(decl) == set(ABI_SYMBOLS) | set(STEREO_ABI_SYMBOLS), set(decl) ^ (set(ABI_SYMBOLS) | set(STEREO_ABI_SYMBOLS))
    lib = C.CDLL(flame_amd.library_path())
    for name in decl:
        assert hasattr(lib, name), name
    nm = subprocess.check_output(["nm", "-D", "--defined-only", flame_amd.library_path()], text=True)
    exported = set(re.findall(r" T (flame_(?:nltgv2|delaunay|stereo|frames)_[a-z_0-9]+)", nm))
    assert set(decl) <= exported


def test_header_is_plain_c(built, tmp_path):
    """The boundary is a C ABI: the header must compile as C99 and as C++."""
    src = tmp_path / "t.c"
    src.write_text('#include "flame_nltgv2.h"\n#include "flame_stereo.h"\n#include "flame_frames.h"\nint main(void){flame_nltgv2_params p; flame_stereo_feature f; (void)p; (void)f;'
                   ' return FLAME_NLTGV2_ABI_VERSION - 1 + (int)sizeof(flame_stereo_feature) - 40;}\n')
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-pedantic", "-I", os.path.join(ROOT, "include"), "-c", str(src),
                           "-o", str(tmp_path / "t.o")])
    subprocess.check_call(["g++", "-std=c++11", "-Wall", "-Werror", "-x", "c++", "-I", os.path.join(ROOT, "include"), "-c", str(src),
                           "-o", str(tmp_path / "t2.o")])


def test_defaults_and_status_strings(built):
    import flame_amd

    lib = flame_amd.load_library()
    assert lib.flame_nltgv2_abi_version() == 7
    p = flame_amd.Params(0, 0, 0, 0, 0, 0)
    lib.flame_nltgv2_default_params(C.byref(p))
    # == nltgv2_l1_graph_regularizer.h:121-129
    assert [round(v, 6) for v in (p.data_factor, p.step_x, p.step_q, p.theta, p.x_min, p.x_max)] == \
        [0.1, 0.001, 125.0, 0.25, 0.0, 10.0]
    for st in range(0, -9, -1):
        assert lib.flame_nltgv2_status_string(st)
    assert b"unknown" in lib.flame_nltgv2_status_string(-99)


@pytest.mark.skipif(HAS_GPU, reason="checks the no-device behaviour")
def test_no_device_fails_loudly(built):
    import flame_amd

    with pytest.raises(flame_amd.NLTGV2Error) as ei:
        flame_amd.Regularizer(0)
    assert ei.value.status == -2  # FLAME_NLTGV2_ERR_NO_DEVICE: no silent CPU path
    from flame_amd.stereo import FeatureTracker

    with pytest.raises(flame_amd.NLTGV2Error) as ei:
        FeatureTracker([525, 0, 320, 0, 525, 240, 0, 0, 1], [1, 0, 0, 0, 1, 0, 0, 0, 1], 640, 480)
    assert ei.value.status == -2


def test_product_package_never_imports_the_checker():
    """oracle/ is test infrastructure: nothing under flame_amd/ may import or load it."""
    pkg = os.path.join(ROOT, "flame_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".hpp", ".cpp", "Makefile")):
                txt = open(os.path.join(dirpath, f), errors="replace").read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", txt, flags=re.M), os.path.join(dirpath, f)
                assert "liboracle" not in txt and "nltgv2_oracle" not in txt, os.path.join(dirpath, f)


def test_pv_residency_table_matches_the_build(tmp_path):
    """The planner's residency of k_persistent_pv (pv_real_waves_per_simd, nltgv2_persistent.hip) is derived from the
    register counts of the instances as built: waves per SIMD = min(512 // VGPRs rounded up to 8,
    800 // (SGPRs rounded up to 16, + 16 for the trap handler), 8) -- the rule tools/residency_probe.hip measured on
    the hardware (profiles/r03_residency.txt), which the runtime's occupancy query does not follow.  Compiles the
    kernels with the compiler's resource report and checks every instance against its row of the table."""
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc")
    src = os.path.join(ROOT, "flame_amd", "csrc", "nltgv2_persistent.hip")
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math",
           "-I", os.path.join(ROOT, "include"), "-I", os.path.join(ROOT, "flame_amd", "csrc"), "-c", src, "-o", str(tmp_path / "k.o"),
           "-Rpass-analysis=kernel-resource-usage"]
    rep = subprocess.run(cmd, capture_output=True, text=True, check=True).stderr
    found = {}
    for m in re.finditer(r"Function Name: (\S+).*?TotalSGPRs: (\d+).*?VGPRs: (\d+).*?ScratchSize \[bytes/lane\]: (\d+)", rep, flags=re.S):
        name, sg, vg, scratch = m.group(1), int(m.group(2)), int(m.group(3)), int(m.group(4))
        k = re.search(r"k_persistent_pvILb([01])ELb([01])ELb([01])E", name)
        if k:
            found[(int(k.group(1)), int(k.group(2)), int(k.group(3)))] = (sg, vg, scratch)
    # (probe, verify, open): plain, verifying, probing, and the open run's instance (flame_nltgv2_run_open)
    assert sorted(found) == [(0, 0, 0), (0, 0, 1), (0, 1, 0), (1, 1, 0)], sorted(found)
    txt = open(src).read()
    assert "return verify_or_probe ? 5 : 7;" in txt, "the table in this test mirrors pv_real_waves_per_simd"
    for (probe, verify, open_run), (sg, vg, scratch) in sorted(found.items()):
        real = min(512 // ((vg + 7) // 8 * 8), 800 // ((sg + 15) // 16 * 16 + 16), 8)
        want = 5 if (probe or verify) else 6 if open_run else 7  # (an open run: graphs of at most 20 patches per CU, nltgv2_run.hip)
        assert real >= want, f"instance probe={probe} verify={verify} open={open_run}: {vg} VGPRs / {sg} SGPRs keep {real} waves per SIMD, the planner assumes {want}"
    # the instances a frame runs on (no probe, no verification) must not spill
    assert found[(0, 0, 0)][2] == 0 and found[(0, 0, 1)][2] == 0
    # k_persistent_pv2 (two half-edges per lane): the planner counts on five waves per SIMD for the plain instance, four for the one
    # with the record verification (pv2_patches_per_cu: 20 / 16 per CU)
    src2 = os.path.join(ROOT, "flame_amd", "csrc", "nltgv2_persistent_pv2.hip")
    cmd2 = [c if c != src else src2 for c in cmd]
    rep2 = subprocess.run(cmd2, capture_output=True, text=True, check=True).stderr
    found2 = {}
    for m in re.finditer(r"Function Name: (\S+).*?TotalSGPRs: (\d+).*?VGPRs: (\d+).*?ScratchSize \[bytes/lane\]: (\d+)", rep2, flags=re.S):
        k = re.search(r"k_persistent_pv2ILb([01])ELb([01])E", m.group(1))
        if k:
            found2[(int(k.group(1)), int(k.group(2)))] = (int(m.group(2)), int(m.group(3)), int(m.group(4)))
    assert sorted(found2) == [(0, 0), (0, 1), (1, 0)], sorted(found2)  # (verify, open)
    for (verify, open_run), (sg, vg, scratch) in found2.items():
        real = min(512 // ((vg + 7) // 8 * 8), 800 // ((sg + 15) // 16 * 16 + 16), 8)
        want = 4 if (verify or open_run) else 5  # (an open run: graphs of at most 14 patches per CU, nltgv2_run.hip)
        assert real >= want and scratch == 0, f"k_persistent_pv2<verify={verify}>: {vg} VGPRs / {sg} SGPRs keep {real} waves per SIMD, the planner assumes {want}"
    txt2 = open(src2).read()
    assert "return n < 20 ? n : 20;" in txt2 and "if (verify) return n < 16 ? n : 16;" in txt2



def test_option_numbers_of_the_python_mirror_match_the_header():
    """flame_amd/regularizer.py names the stable options of include/flame_nltgv2.h (and a few of the experimental range of
    flame_amd/csrc/flame_nltgv2_test_options.h) by number: every OPT_* constant of the mirror must be the header's value."""
    import re

    from flame_amd import regularizer

    text = open(os.path.join(ROOT, "include", "flame_nltgv2.h")).read()
    text += open(os.path.join(ROOT, "flame_amd", "csrc", "flame_nltgv2_test_options.h")).read()
    header = {m.group(1): int(m.group(2)) for m in re.finditer(r"FLAME_NLTGV2_(OPT_[A-Z0-9_]+)\s*=\s*(\d+)", text)}
    mirror = {k: v for k, v in vars(regularizer).items() if k.startswith("OPT_") and isinstance(v, int)}
    assert len(mirror) >= 9 and "OPT_MESH_STATE" in mirror
    for name, value in mirror.items():
        assert header.get(name) == value, (name, value, header.get(name))
