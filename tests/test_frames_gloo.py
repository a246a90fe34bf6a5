"""N>1 path on CPU: world_size-2 `gloo` processes run the frame sharding + result gather that
bench.py uses with RCCL on the GPU node.  The per-frame "solver result" here comes from the CPU
checker (tests may use it); what is under test is the host logic: who owns which frame, the ragged
gather layout, and that every rank ends up with every frame's vector."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tests.conftest import ROOT


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_frames, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from flame_amd import synth
        from flame_amd.frames import IdepthGather, shard_frames
        from oracle import capi as oracle

        mine = shard_frames(n_frames, world, rank)
        graphs = {f: synth.make_graph("320x240", seed=900 + f) for f in mine}
        # ragged on purpose: drop a few vertices' worth of length on odd frames
        sizes = [graphs[f]["V"] for f in mine]
        ig = IdepthGather(dist, sizes, n_frames, torch.device("cpu"))
        for i, f in enumerate(mine):
            oracle.run(graphs[f], 10)
            ig.local_row(i)[: graphs[f]["V"]] = torch.from_numpy(graphs[f]["x"] * np.float32(2.0))
        ig.gather()
        ok = True
        for f in range(n_frames):
            ref = synth.make_graph("320x240", seed=900 + f)
            oracle.run(ref, 10)
            got = ig.frame(f).numpy()
            ok = ok and got.shape[0] == ref["V"] and np.array_equal(got, ref["x"] * np.float32(2.0))
        q.put((rank, mine, ok))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_frames", [2, 3])
def test_world2_gloo_frame_sharding_and_gather(n_frames):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_frames, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    owned = sorted(f for _, mine, _ in res for f in mine)
    assert owned == list(range(n_frames))  # every frame solved exactly once
    assert all(ok for _, _, ok in res)


def test_shard_frames_properties():
    from flame_amd.frames import frames_per_rank, shard_frames

    for n in range(0, 20):
        for w in range(1, 9):
            all_ids = [f for r in range(w) for f in shard_frames(n, w, r)]
            assert all_ids == list(range(n))
            c = frames_per_rank(n, w)
            assert max(c) - min(c) <= 1
    with pytest.raises(ValueError):
        shard_frames(4, 2, 2)
