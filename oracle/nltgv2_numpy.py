"""Second, independently written restatement of the reference solver in numpy float32
(test infrastructure; PARITY UNPINNED like nltgv2_oracle.c).

Written from the maths of nltgv2_l1_graph_regularizer.cc:33-174 rather than from the C file: the
edge scatter of primalStep (cc:120-142) is expressed as one in-order `np.add.at` over an
interleaved (vertex, value) stream, which reproduces the reference's per-vertex accumulation order
exactly, so the C and numpy restatements must agree BIT FOR BIT (tests/test_oracle.py).
"""
from __future__ import annotations

import numpy as np

F = np.float32


def _prox_conj(q):
    # h:171-176: q / max(1,|q|) with fast_abs(r) = (r > 0) ? r : -r
    absq = np.where(q > 0, q, -q)
    return (q / np.where(absq > 1, absq, F(1))).astype(F)


def step(g: dict, p: dict) -> None:
    sx, sq, th, lam = F(p["step_x"]), F(p["step_q"]), F(p["theta"]), F(p["data_factor"])
    xmin, xmax = F(p["x_min"]), F(p["x_max"])
    i, j = g["src"], g["dst"]
    al, be = g["alpha"], g["beta"]
    dx = (g["pos"][i, 0] - g["pos"][j, 0]).astype(F)
    dy = (g["pos"][i, 1] - g["pos"][j, 1]).astype(F)

    g["x_prev"][:] = g["x"]
    g["w1_prev"][:] = g["w1"]
    g["w2_prev"][:] = g["w2"]

    # dualStep cc:89-114
    xb, w1b, w2b = g["x_bar"], g["w1_bar"], g["w2_bar"]
    K1 = al * (xb[i] - xb[j])
    K1 = K1 - (al * dx) * w1b[i]
    K1 = K1 - (al * dy) * w2b[i]
    g["q1"][:] = _prox_conj(g["q1"] + sq * K1)
    g["q2"][:] = _prox_conj(g["q2"] + sq * (be * (w1b[i] - w1b[j])))
    g["q3"][:] = _prox_conj(g["q3"] + sq * (be * (w2b[i] - w2b[j])))

    # primalStep cc:116-154: per-edge sequence of in-place updates, kept in order
    t1 = (g["q1"] * sx) * al
    t2 = (g["q2"] * sx) * be
    t3 = (g["q3"] * sx) * be
    E = i.shape[0]
    idx = np.empty(2 * E, dtype=np.int64)
    idx[0::2], idx[1::2] = i, j
    val = np.empty(2 * E, dtype=F)
    val[0::2], val[1::2] = -t1, t1
    np.add.at(g["x"], idx, val)
    idx3 = np.empty(3 * E, dtype=np.int64)
    idx3[0::3], idx3[1::3], idx3[2::3] = i, i, j
    val3 = np.empty(3 * E, dtype=F)
    val3[0::3], val3[1::3], val3[2::3] = t1 * dx, -t2, t2
    np.add.at(g["w1"], idx3, val3)
    val3[0::3], val3[1::3], val3[2::3] = t1 * dy, -t3, t3
    np.add.at(g["w2"], idx3, val3)

    x, d = g["x"], g["data_term"]
    thr = sx * (lam * g["data_weight"])
    diff = x - d
    nx = np.where(diff > thr, x - thr, np.where(diff < -thr, x + thr, d)).astype(F)
    nx = np.where(nx < xmin, xmin, nx)
    nx = np.where(nx > xmax, xmax, nx)
    g["x"][:] = nx

    # extraGradientStep cc:156-174
    nb = g["x"] + th * (g["x"] - g["x_prev"])
    nb = np.where(nb < xmin, xmin, nb)
    nb = np.where(nb > xmax, xmax, nb)
    g["x_bar"][:] = nb
    g["w1_bar"][:] = g["w1"] + th * (g["w1"] - g["w1_prev"])
    g["w2_bar"][:] = g["w2"] + th * (g["w2"] - g["w2_prev"])


def run(g: dict, n_iters: int, p: dict) -> None:
    for _ in range(n_iters):
        step(g, p)
