"""Host logic without a GPU: the SELL-64 packing produced by the C-ABI's pack probe is structurally
sound, and a numpy EMULATION of the fused sweep over that packed layout (one vertex per lane, slots
in ascending edge id, private q copy per half-edge) reproduces the checker bit for bit -- i.e. the
vertex-gather re-association the GPU kernels use is exact, not approximate."""
import os

import numpy as np
import pytest

from flame_amd import regularizer, synth
from oracle import capi as oracle
from tests.conftest import ROOT
from tests.helpers import assert_state_equal, load_golden, random_graph

ROLE = np.uint32(0x80000000)
F = np.float32


def emulate_fused(g, probe, n_iters, p):
    """numpy mirror of k_fused_step (flame_amd/csrc/nltgv2_kernels.hip): per slot k, all vertices at
    once; each vertex adds its k-th incident half-edge's contribution -> ascending edge id per vertex."""
    perm, srow = probe["perm"], probe["slice_row"]
    nbr_role, redge = probe["rec_nbr"], probe["rec_edge"]
    n_packed = perm.shape[0]
    live = perm >= 0
    o = np.where(live, perm, 0)
    x = np.where(live, g["x"][o], 0).astype(F)
    w1 = np.where(live, g["w1"][o], 0).astype(F)
    w2 = np.where(live, g["w2"][o], 0).astype(F)
    xb = np.where(live, g["x_bar"][o], 0).astype(F)
    w1b = np.where(live, g["w1_bar"][o], 0).astype(F)
    w2b = np.where(live, g["w2_bar"][o], 0).astype(F)
    data = np.where(live, g["data_term"][o], 0).astype(F)
    wgt = np.where(live, g["data_weight"][o], 0).astype(F)
    has = redge >= 0
    e = np.where(has, redge, 0)
    alpha = np.where(has, g["alpha"][e], 0).astype(F)
    beta = np.where(has, g["beta"][e], 0).astype(F)
    dx = np.where(has, g["pos"][g["src"][e], 0] - g["pos"][g["dst"][e], 0], 0).astype(F)
    dy = np.where(has, g["pos"][g["src"][e], 1] - g["pos"][g["dst"][e], 1], 0).astype(F)
    q = [np.where(has, g[k][e], 0).astype(F) for k in ("q1", "q2", "q3")]
    nbr = (nbr_role & ~ROLE).astype(np.int64)
    is_t = (nbr_role & ROLE) != 0
    sx, sq, th, lam = F(p["step_x"]), F(p["step_q"]), F(p["theta"]), F(p["data_factor"])
    xmin, xmax = F(p["x_min"]), F(p["x_max"])
    n_slices = srow.shape[0] - 1
    width = np.diff(srow)
    lane = np.arange(64)
    xp = x.copy()
    for _ in range(n_iters):
        xp, w1p, w2p = x.copy(), w1.copy(), w2.copy()
        nx, nw1, nw2 = x.copy(), w1.copy(), w2.copy()
        for k in range(int(width.max()) if n_slices else 0):
            sl = np.nonzero(width > k)[0]
            slot = ((srow[sl] + k)[:, None] * 64 + lane[None, :]).ravel()
            v = (sl[:, None] * 64 + lane[None, :]).ravel()
            m = has[slot]
            slot, v = slot[m], v[m]
            n = nbr[slot]
            t = is_t[slot]
            i = np.where(t, n, v)
            j = np.where(t, v, n)
            K1 = alpha[slot] * (xb[i] - xb[j])
            K1 = K1 - alpha[slot] * dx[slot] * w1b[i]
            K1 = K1 - alpha[slot] * dy[slot] * w2b[i]
            q1 = np.clip(q[0][slot] + sq * K1, F(-1), F(1)).astype(F)
            q2 = np.clip(q[1][slot] + sq * (beta[slot] * (w1b[i] - w1b[j])), F(-1), F(1)).astype(F)
            q3 = np.clip(q[2][slot] + sq * (beta[slot] * (w2b[i] - w2b[j])), F(-1), F(1)).astype(F)
            q[0][slot], q[1][slot], q[2][slot] = q1, q2, q3
            t1 = q1 * sx * alpha[slot]
            t2 = q2 * sx * beta[slot]
            t3 = q3 * sx * beta[slot]
            nx[v] = np.where(t, nx[v] + t1, nx[v] - t1)
            nw1[v] = np.where(t, nw1[v] + t2, (nw1[v] + t1 * dx[slot]) - t2)
            nw2[v] = np.where(t, nw2[v] + t3, (nw2[v] + t1 * dy[slot]) - t3)
        thr = sx * (lam * wgt)
        diff = nx - data
        px = np.where(diff > thr, nx - thr, np.where(diff < -thr, nx + thr, data)).astype(F)
        px = np.where(px < xmin, xmin, px)
        px = np.where(px > xmax, xmax, px)
        x, w1, w2 = px.astype(F), nw1, nw2
        b = x + th * (x - xp)
        b = np.where(b < xmin, xmin, b)
        xb = np.where(b > xmax, xmax, b).astype(F)
        w1b = (w1 + th * (w1 - w1p)).astype(F)
        w2b = (w2 + th * (w2 - w2p)).astype(F)
    out = {k: np.zeros(g["V"], F) for k in ("x", "w1", "w2", "x_bar", "w1_bar", "w2_bar")}
    for k, arr in (("x", x), ("w1", w1), ("w2", w2), ("x_bar", xb), ("w1_bar", w1b), ("w2_bar", w2b)):
        out[k][perm[live]] = arr[live]
    src_side = has & ~is_t
    for k, arr in zip(("q1", "q2", "q3"), q):
        out[k] = np.zeros(g["E"], F)
        out[k][redge[src_side]] = arr[src_side]
    # both private copies of every edge must be identical
    tgt_side = has & is_t
    for arr in q:
        a = np.zeros(g["E"], F); a[redge[src_side]] = arr[src_side]
        b2 = np.zeros(g["E"], F); b2[redge[tgt_side]] = arr[tgt_side]
        assert np.array_equal(a, b2)
    return out


def check_structure(g, pr):
    V, E = g["V"], g["E"]
    perm, srow, redge, nbr_role = pr["perm"], pr["slice_row"], pr["rec_edge"], pr["rec_nbr"]
    assert pr["n_slices"] == (V + 63) // 64
    live = perm[perm >= 0]
    assert np.array_equal(np.sort(live), np.arange(V))  # a permutation of the vertices
    assert np.all(perm[:V] >= 0) and np.all(perm[V:] == -1)  # padding only at the very end
    assert srow[0] == 0 and np.all(np.diff(srow) >= 0) and srow[-1] == pr["rows"]
    e = redge[redge >= 0]
    assert e.size == 2 * E and np.array_equal(np.bincount(e, minlength=E), np.full(E, 2))  # each edge twice
    iperm = np.empty(V, np.int64)
    iperm[perm[:V]] = np.arange(V)
    slots = np.nonzero(redge >= 0)[0]
    for s in slots[:: max(1, slots.size // 5000)]:  # sampled deep check
        row, lane = divmod(int(s), 64)
        sl = int(np.searchsorted(srow, row, side="right") - 1)
        v = perm[sl * 64 + lane]
        ed = redge[s]
        is_t = bool(nbr_role[s] & ROLE)
        other = int(nbr_role[s] & ~ROLE)
        if is_t:
            assert g["dst"][ed] == v and iperm[g["src"][ed]] == other
        else:
            assert g["src"][ed] == v and iperm[g["dst"][ed]] == other
    # ascending edge id down the slots of every lane
    for sl in range(pr["n_slices"]):
        blk = redge[srow[sl] * 64: srow[sl + 1] * 64].reshape(-1, 64).astype(np.int64)
        if blk.shape[0] < 2:
            continue
        filled = blk >= 0
        assert np.all(filled[:-1] | ~filled[1:])  # no holes above a filled slot
        asc = (blk[1:] > blk[:-1]) | ~filled[1:]
        assert np.all(asc)


@pytest.mark.parametrize("make", [
    lambda: load_golden("cfg1_320x240_s1234")[0],
    lambda: random_graph(700, 2500, seed=1),
    lambda: random_graph(129, 400, seed=4),
    lambda: synth.concat_graphs([synth.make_graph("320x240", seed=50 + i) for i in range(3)]),
])
def test_pack_structure_and_emulated_sweep(built, make):
    g = make()
    pr = regularizer.pack_probe(g)
    check_structure(g, pr)
    ref = synth.copy_graph(g)
    oracle.run(ref, 12)
    out = emulate_fused(g, pr, 12, oracle.DEFAULT_PARAMS)
    assert_state_equal(out, ref, what="emulated fused sweep")


def test_pack_rejects_bad_graphs(built):
    g = random_graph(50, 100, seed=2)
    b = synth.copy_graph(g)
    b["src"][3] = 50
    with pytest.raises(regularizer.NLTGV2Error):
        regularizer.pack_probe(b)
    b = synth.copy_graph(g)
    b["dst"][3] = b["src"][3]
    with pytest.raises(regularizer.NLTGV2Error):
        regularizer.pack_probe(b)
    b = synth.copy_graph(g)
    b["pos"][7, 0] = np.nan
    with pytest.raises(regularizer.NLTGV2Error):
        regularizer.pack_probe(b)


def test_batch_frames_stay_contiguous(built):
    """Connected components (= frames of a batch) are packed one after the other even though they
    overlap in image coordinates."""
    frames = [synth.make_graph("320x240", seed=70 + i) for i in range(4)]
    g = synth.concat_graphs(frames)
    pr = regularizer.pack_probe(g)
    V1 = frames[0]["V"]
    frame_of = pr["perm"][: g["V"]] // V1
    assert np.all(np.diff(frame_of) >= 0)


def test_wg_layout_replay(built, tmp_path):
    """Layout (E) of nltgv2_pack.hpp (patch-per-workgroup rows): tests/cpp/wg_layout_test.cc replays the kernel's data
    movement on the CPU with the device code's visibility rules and compares with the checker bit for bit."""
    import subprocess

    exe = str(tmp_path / "wg_layout_test")
    subprocess.check_call([
        "g++", "-std=c++17", "-O1", "-ffp-contract=off", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(ROOT, "include"),
        "-I", os.path.join(ROOT, "flame_amd", "csrc"), os.path.join(ROOT, "tests", "cpp", "wg_layout_test.cc"), "-o", exe,
        "-L", os.path.join(ROOT, "oracle"), "-loracle_nltgv2", f"-Wl,-rpath,{os.path.join(ROOT, 'oracle')}"])
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "all ok" in r.stdout, r.stdout + r.stderr
