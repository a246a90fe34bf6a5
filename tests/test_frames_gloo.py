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


class _FakeSolver:
    """Stands in for a Regularizer in the host-logic test of IdepthGather.settle(): run() fills the send row the way the
    solver's export does, except that a 'timed-out' run leaves the row alone until sync() redoes it (what a persistent run
    that leaves through an expired wait does: no epilogue export; flame_nltgv2_sync replays and re-exports)."""

    def __init__(self):
        self.recovered, self.pending = 0, None

    def run(self, row, value, time_out=False):
        if time_out:
            self.pending = (row, value)
        else:
            row[:] = value

    def sync(self):
        if self.pending is not None:
            row, value = self.pending
            row[:] = value          # the replay's export
            self.recovered += 1
            self.pending = None

    def info(self):
        return {"timeouts_recovered": self.recovered, "torn_records_detected": 0}


def _settle_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from flame_amd.frames import IdepthGather

        ig = IdepthGather(dist, [5], world, torch.device("cpu"))
        s = _FakeSolver()
        out = []
        for step in range(3):
            # step 1: rank 1's run "times out" -- its row still holds step 0's values when the gather reads it
            s.run(ig.local_row(0), float(10 * step + rank), time_out=(step == 1 and rank == 1))
            ig.gather(async_op=True, regs=[s])
            stale = [float(ig.frame(f)[0]) for f in range(world)] if step == 1 else None
            redo = ig.settle([s])
            out.append((redo, [float(ig.frame(f)[0]) for f in range(world)], stale))
        q.put((rank, out))
    finally:
        dist.destroy_process_group()


def test_world2_gloo_settle_regathers_after_a_replayed_run():
    """ADVICE r02 (frame_gather.hpp:9): a gather enqueued behind an unchecked run can carry a stale row; settle() must notice
    on EVERY rank (the rank that replayed and the ones that did not) and gather again."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_settle_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank in (0, 1):
        (r0, v0, _), (r1, v1, stale), (r2, v2, _) = res[rank]
        assert (r0, r1, r2) == (0, 1, 0), (rank, r0, r1, r2)
        assert v0 == [0.0, 1.0] and v2 == [20.0, 21.0]
        assert stale[0] == 10.0 and stale[1] != 11.0  # what the first gather of step 1 delivered: rank 1's row as the run found it
        assert v1 == [10.0, 11.0]       # ... and what settle() left


def test_overlap_verdict_reports_a_gather_that_waits_for_the_runs():
    """IdepthGather.overlap_verdict: round 4's two measurements -- 0.35 ms per step with RCCL's stream on the solver's hardware queue,
    0.21 beside it, 0.19-0.21 without a gather -- and what the host is told."""
    import warnings

    from flame_amd.frames import IdepthGather

    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        good = IdepthGather.overlap_verdict(0.1908, 0.2094)
        assert good["overlaps"] and not w and abs(good["gather_tax"] - 0.0975) < 1e-3
        bad = IdepthGather.overlap_verdict(0.21, 0.35)
        assert not bad["overlaps"] and len(w) == 1 and "GPU_MAX_HW_QUEUES=8" in str(w[0].message)
