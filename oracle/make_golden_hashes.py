"""Generates tests/golden/config_hashes.json: for the BASELINE sizes too large for full fixtures (1280x720, 1920x1080;
SURVEY.md 8(c)), SHA-256 of every state array and the two costs after 200 steps of the C restatement
(oracle/nltgv2_oracle.c), cross-checked bit-exact against the numpy restatement, on the seeded synthetic graph that
flame_amd.synth builds with the library's own triangulator.  Usage:  python -m oracle.make_golden_hashes

Like the .npz fixtures these freeze the CHECKER (parity unpinned: the reference's translation unit cannot be built
here); they let the GPU box verify a BASELINE-size run against numbers produced and reviewed in the build container."""
from __future__ import annotations

import hashlib
import json
import os

import numpy as np

from flame_amd import synth
from oracle import capi, nltgv2_numpy

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "config_hashes.json")
KEYS = ("x", "w1", "w2", "x_bar", "w1_bar", "w2_bar", "q1", "q2", "q3")


def digest(a: np.ndarray) -> str:
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def entry(config: str, seed: int, iters: int, cross_check: bool) -> dict:
    g = synth.make_graph(config, seed)
    inputs = {k: digest(g[k]) for k in ("pos", "data_term", "src", "dst", "alpha")}
    a = synth.copy_graph(g)
    assert capi.run(a, iters) == 0
    if cross_check:
        b = synth.copy_graph(g)
        nltgv2_numpy.run(b, iters, capi.DEFAULT_PARAMS)
        for k in KEYS:
            assert np.array_equal(a[k], b[k]), (config, k)
    sm, dc = capi.costs(a)
    return {"config": config, "seed": seed, "iters": iters, "V": int(g["V"]), "E": int(g["E"]), "inputs": inputs,
            "state": {k: digest(a[k]) for k in KEYS},
            "costs_f32_bits": [int(np.float32(sm).view(np.uint32)), int(np.float32(dc).view(np.uint32))]}


def main():
    out = [entry("1280x720", 1234, 200, True), entry("1920x1080", 1234, 200, True)]
    with open(OUT, "w") as f:
        json.dump(out, f, indent=1)
    print(OUT, [(e["config"], e["V"], e["E"]) for e in out])


if __name__ == "__main__":
    main()
