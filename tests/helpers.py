"""Shared helpers of the test-suite (tests may use oracle/)."""
from __future__ import annotations

import os

import numpy as np

from flame_amd import synth

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
INPUT_KEYS = ("pos", "data_term", "data_weight", "src", "dst", "alpha", "beta")
OUT_KEYS = ("x", "w1", "w2", "x_bar", "w1_bar", "w2_bar", "q1", "q2", "q3")


def load_golden(name: str):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    edges = np.stack([z["src"], z["dst"]], axis=1)
    g = synth.assemble_graph(z["pos"], z["data_term"], edges, weight=z["data_weight"])
    g["alpha"] = z["alpha"].copy()
    g["beta"] = z["beta"].copy()
    return g, z


def assert_state_equal(a: dict, b: dict, keys=OUT_KEYS, what=""):
    for k in keys:
        if not np.array_equal(a[k], b[k]):
            d = np.abs(a[k].astype(np.float64) - b[k].astype(np.float64))
            raise AssertionError(f"{what}: {k} differs: max {d.max():.3e} at {int(d.argmax())}, "
                                 f"{int((d > 0).sum())}/{d.size} elements")


def rms(a, b):
    d = a.astype(np.float64) - b.astype(np.float64)
    return float(np.sqrt(np.mean(d * d))) if d.size else 0.0


def random_graph(V: int, E: int, seed: int, width=100.0):
    """Random simple graph (no self loops / duplicates) with random orientation and weights."""
    rng = np.random.default_rng(seed)
    pos = (rng.random((V, 2)) * width).astype(np.float32)
    data = (0.5 + rng.random(V)).astype(np.float32)
    seen, edges = set(), []
    tries = 0
    while len(edges) < E and tries < 50 * E + 100:
        tries += 1
        i, j = int(rng.integers(V)), int(rng.integers(V))
        if i == j or (min(i, j), max(i, j)) in seen:
            continue
        if np.all(pos[i] == pos[j]):
            continue
        seen.add((min(i, j), max(i, j)))
        edges.append((i, j))
    edges = np.array(edges, dtype=np.int32).reshape(-1, 2)
    g = synth.assemble_graph(pos, data, edges, weight=(0.5 + rng.random(V)).astype(np.float32))
    if len(edges):
        g["beta"] = (0.5 + rng.random(len(edges))).astype(np.float32)
    return g
