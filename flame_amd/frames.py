"""Frame-level data parallelism: independent frames (one Delaunay graph each) are sharded across
ranks -- one process per GPU -- with NO collective on the solve path; the only exchange is the result
gather of `x * graph_scale` (what Flame::update reads back, /root/reference/src/flame/flame.cc:372-380),
done with torch.distributed (backend "nccl" = RCCL over xGMI on the GPU node, "gloo" in CPU tests).

A single graph is never split across GPUs: at <= 11 MB per step and two neighbour exchanges per step
the xGMI latency would dominate (DESIGN.md "Multi-GPU").
"""
from __future__ import annotations

from typing import List, Sequence

import torch


def shard_frames(n_frames: int, world: int, rank: int) -> List[int]:
    """Frame ids owned by `rank`: contiguous blocks, sizes differing by at most one."""
    if world <= 0 or not (0 <= rank < world) or n_frames < 0:
        raise ValueError("bad sharding arguments")
    base, extra = divmod(n_frames, world)
    start = rank * base + min(rank, extra)
    return list(range(start, start + base + (1 if rank < extra else 0)))


def frames_per_rank(n_frames: int, world: int) -> List[int]:
    return [len(shard_frames(n_frames, world, r)) for r in range(world)]


class IdepthGather:
    """Gathers every rank's per-frame inverse-depth vectors (ragged: V differs per frame) to all ranks.

    Layout: each rank contributes a [slots, vmax] float32 block (slots = max frames per rank, vmax =
    max vertex count over all frames, zero padded), so ONE all_gather_into_tensor of a fixed shape moves
    everything -- at ~34 KB per 640x480 frame this is a pure-latency collective, one call per step.
    """

    def __init__(self, dist, sizes_local: Sequence[int], n_frames: int, device, stream=None):
        """`stream`: the torch stream the local solvers run on (what Regularizer.set_stream was given).  The waits of local_row() and the
        collective of gather() are then ordered against THAT stream whatever torch's current stream is when they are called.  Without
        it they are ordered against torch's current stream -- the caller's job to make that the solver's: a wait that lands on the
        default stream orders nothing the solver does, and its barrier packets occupy one more of the process's hardware queues
        (4 by default: a step of 0.28-0.35 instead of 0.20 ms at 640x480, tools/overlap_probe.py)."""
        self.dist = dist
        self.stream = stream
        self._side = None  # (set below: the stream the collective is issued from when the solvers hand their runs over themselves)
        self.world = dist.get_world_size()
        self.rank = dist.get_rank()
        self.n_frames = n_frames
        self.slots = max(frames_per_rank(n_frames, self.world)) if n_frames else 0
        self.my_frames = shard_frames(n_frames, self.world, self.rank)
        if len(sizes_local) != len(self.my_frames):
            raise ValueError("one size per local frame expected")
        # Staged mode: the send rows live on a GPU but the backend moves host memory only (gloo has no device all_gather).
        # gather() then copies the local rows to pinned host memory behind the solver's run, the collective runs on the host
        # copies and `gathered` / frame() are host tensors.  This is how two ranks drive real solvers on ONE device (RCCL refuses
        # two ranks per device): tests/test_frames_world2_gpu.py, bench.py with FLAME_BENCH_BACKEND=gloo.
        self._staged = torch.device(device).type == "cuda" and str(dist.get_backend()) == "gloo"
        cdev = torch.device("cpu") if self._staged else device
        # agree on vmax and publish every frame's true length
        sizes = torch.zeros(self.world * max(self.slots, 1), dtype=torch.int64, device=cdev)
        mine = torch.zeros(max(self.slots, 1), dtype=torch.int64, device=cdev)
        for i, v in enumerate(sizes_local):
            mine[i] = int(v)
        dist.all_gather_into_tensor(sizes, mine)
        self.sizes = sizes.cpu().view(self.world, max(self.slots, 1))
        self.vmax = int(self.sizes.max().item()) if n_frames else 0
        # double buffered: the gather of step k runs (on the collective's stream) while the solver already
        # works on step k+1 and exports into the other buffer
        shape_l = (max(self.slots, 1), max(self.vmax, 1))
        shape_g = (self.world * max(self.slots, 1), max(self.vmax, 1))
        self._local = [torch.zeros(shape_l, dtype=torch.float32, device=device) for _ in range(2)]
        self._gathered = [torch.empty(shape_g, dtype=torch.float32, device=cdev) for _ in range(2)]
        self._h_local = [torch.zeros(shape_l, dtype=torch.float32).pin_memory() for _ in range(2)] if self._staged else None
        self._work = [None, None]
        self._done = [None, None]  # side-stream mode: the event behind the collective that used buffer k
        self._cur = 0
        # Side-stream mode (round 5): with the solvers' stream known and a device backend, gather(regs=...) issues the collective from a
        # stream of its own that waits for the solvers' runs through Regularizer.stream_wait_run -- the run's launch carries that event --
        # and the rows are given back to the solvers by a HOST wait for the collective two steps back.  The solver's in-order queue
        # then holds its launches only: the event torch records for the collective and the wait before a row is reused were two
        # operations per step between two solver launches, 10 us of a 185 us step.
        import os

        if stream is not None and not self._staged and torch.device(device).type == "cuda" and os.environ.get("FLAME_GATHER_SIDE_STREAM", "1") != "0":
            self._side = torch.cuda.Stream(device=device)  # (FLAME_GATHER_SIDE_STREAM=0: the collective ordered by torch's own events on the solver's stream)
            self._done_ev = [torch.cuda.Event(), torch.cuda.Event()]
        self.local = self._local[0]
        self.gathered = self._gathered[0]

    def _on_stream(self):
        import contextlib

        return torch.cuda.stream(self.stream) if self.stream is not None else contextlib.nullcontext()

    def _wait_work(self, w) -> None:
        """Orders a collective's result for its two readers: the SOLVER's stream (which re-uses the rows) and the CALLER's current
        stream (which reads `gathered` / frame()).  Work.wait() blocks only the stream that is current when it is called; with
        `stream=` set that used to be the solver's alone, and a `frame(i).cpu()` on the default stream was ordered behind the
        collective by timing only (advisor, round 5)."""
        w.wait()  # the caller's current stream (a host backend: the host)
        if self.stream is not None and not self._staged and torch.cuda.current_stream(self.stream.device) != self.stream:
            with torch.cuda.stream(self.stream):
                w.wait()

    def local_row(self, i: int) -> torch.Tensor:
        """Device row the solver of local frame i writes its x*scale into (first V entries).  Valid until
        the next gather(); the solver's stream waits for the collective that last read this buffer."""
        if self._done[self._cur] is not None:  # (the host waits: the row is free before the next run is even enqueued)
            self._done[self._cur].synchronize()
            self._done[self._cur] = None
        if self._work[self._cur] is not None:
            with self._on_stream():
                self._work[self._cur].wait()
            self._work[self._cur] = None
        return self._local[self._cur][i]

    @staticmethod
    def _replays(regs):
        out = []
        for r in regs:
            i = r.info()
            out.append(i["timeouts_recovered"] + i["torn_records_detected"])
        return out

    def gather(self, async_op: bool = False, regs=None) -> None:
        """all_gather of the current local buffer.  async_op=True returns at once; frame() / wait() / settle() or
        the next reuse of the buffer completes it.  `regs` (the local solvers whose runs feed the rows): remembered
        for settle()."""
        k = self._cur
        self._replays_at_gather = self._replays(regs) if regs is not None else None
        send = self._local[k]
        if self._staged:  # device rows -> pinned host rows, ordered behind the solver on the current stream, then a host collective
            with self._on_stream():
                self._h_local[k].copy_(self._local[k], non_blocking=True)
                torch.cuda.current_stream(self._local[k].device).synchronize()
            send = self._h_local[k]
        if self._side is not None and regs is not None and not self._staged:
            for r in regs:  # the side stream waits for every solver's run (at no cost to the solvers' streams)
                r.stream_wait_run(self._side.cuda_stream)
            with torch.cuda.stream(self._side):
                w = self.dist.all_gather_into_tensor(self._gathered[k], send, async_op=True)
                w.wait()  # (the side stream waits for the collective's own stream ...)
                self._done_ev[k].record(self._side)  # (... and says when it is over)
            self._done[k] = self._done_ev[k]
            self._work[k] = None
            if not async_op:
                self._done[k].synchronize()
                self._done[k] = None
        else:
            with self._on_stream():  # (the collective starts behind what the solver's stream holds now: the run that exports the rows)
                w = self.dist.all_gather_into_tensor(self._gathered[k], send, async_op=True)
            self._work[k] = w
            if not async_op:  # complete for the solver's stream AND for the stream the caller reads `gathered` on
                self._wait_work(w)
                self._work[k] = None
        self.gathered = self._gathered[k]
        self._last = k
        self._cur = 1 - k
        self.local = self._local[self._cur]

    def check_overlap(self, reg, params, iters: int = 200, steps: int = 24) -> dict:
        """Does the collective run BESIDE the solver, as the pipelined frame loop assumes?  Measured on the streams the frame loop really
        uses: `steps` frames of run_async alone, then the same with the export target set and an asynchronous gather behind every run; a
        step that grows by more than a third means the gather waits for the runs around it: a RuntimeWarning says so and the figures are
        kept in `self.overlap`.  Two ways to get there, both met: (1) the gather's waits (local_row) issued on a stream that is not the
        solver's -- torch's default stream, when neither `stream=` was given nor the caller is inside the solver's stream context: the
        waits order nothing and their packets take a hardware queue of their own (rounds 4-5: 0.28-0.35 instead of 0.20 ms per step with
        the runtime's default of 4 queues); (2) the collective's stream sharing the solver's in-order hardware queue (the runtime
        multiplexes a process's streams onto GPU_MAX_HW_QUEUES queues per priority level; a high-priority solver stream, as the context's
        own is, has a pool to itself).  Collective: every rank calls it (from the solver's stream context, or with the gather's `stream`
        set).  A staged (host) backend has nothing to measure."""
        import time

        if self._staged or self.n_frames == 0:
            self.overlap = {"measured": False}
            return self.overlap
        dev = self._local[0].device

        def loop(with_gather: bool) -> float:
            reg.sync()
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            for _ in range(steps):
                if with_gather:
                    reg.set_export_target(self.local_row(0).data_ptr(), 1.0)
                reg.run_async(params, iters)
                if with_gather:
                    self.gather(async_op=True, regs=[reg])
            reg.sync()
            self.wait()
            torch.cuda.synchronize(dev)
            return (time.perf_counter() - t0) / steps * 1e3

        loop(True)  # (first launches, the communicator's first collective)
        alone, beside = loop(False), loop(True)
        reg.set_export_target(None)
        self.overlap = self.overlap_verdict(alone, beside)
        return self.overlap

    @staticmethod
    def overlap_verdict(step_ms_alone: float, step_ms_with_gather: float) -> dict:
        """The figures of check_overlap and what they mean: a step that grows by more than a third with the gather behind every run is
        a gather that waits for the runs around it (measured: 0.35 against 0.21 ms when its waits sat on torch's default stream with
        four hardware queues, 0.20 against 0.185 when they were ordered on the solver's stream) -- reported as a RuntimeWarning with the
        remedies."""
        import warnings

        ok = bool(step_ms_with_gather < 1.35 * step_ms_alone)
        if not ok:
            warnings.warn("flame_amd.frames: the result gather does not overlap the solver (%.3f ms per step with the gather behind every run, %.3f without): "
                          "give IdepthGather the solver's stream (stream=...), and if it has it, the collective's stream shares the solver's hardware queue: "
                          "use a high-priority solver stream or start the process with GPU_MAX_HW_QUEUES=8"
                          % (step_ms_with_gather, step_ms_alone), RuntimeWarning)
        return {"measured": True, "overlaps": ok, "step_ms_alone": round(step_ms_alone, 4), "step_ms_with_gather": round(step_ms_with_gather, 4),
                "gather_tax": round(step_ms_with_gather / step_ms_alone - 1.0, 4)}

    def settle(self, regs) -> int:
        """Completes the gather and makes sure it carried rows of runs that really finished.  `gather()` right behind
        `run_async()` reads the export rows of unchecked runs: a persistent run whose neighbour wait expires leaves
        before its epilogue writes the row, and `Regularizer.sync()` only then takes it back, redoes the steps and
        re-exports -- after the collective has delivered the previous frame's row.  settle() syncs every local solver
        and, if any rank had to redo a run since the gather was issued, gathers once more from the re-exported rows
        (collectively: the ranks agree through one small all_reduce).  Returns the number of re-gathers (0 or 1)."""
        before = getattr(self, "_replays_at_gather", None) or self._replays(regs)
        for r in regs:
            r.sync()
        redo = 1 if self._replays(regs) != before else 0
        self._replays_at_gather = None
        self.wait()
        if self.world > 1:
            t = torch.tensor([redo], dtype=torch.int32, device="cpu" if self._staged else self._local[0].device)
            self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
            redo = int(t.item())
        if redo:
            self._cur = self._last  # the rows the solvers have just re-exported into
            self.gather(async_op=False)
        return redo

    def wait(self) -> None:
        """Completes the outstanding collectives for the caller's current stream and for the solver's stream (see frame())."""
        for k in (0, 1):
            if self._done[k] is not None:
                self._done[k].synchronize()
                self._done[k] = None
            if self._work[k] is not None:
                self._wait_work(self._work[k])
                self._work[k] = None

    def frame(self, frame_id: int) -> torch.Tensor:
        """x*scale of global frame `frame_id` from the most recent gather().  The returned view is valid on the stream that is CURRENT
        when frame() is called (and on the solver's stream): wait() orders both behind the collective -- by a host wait in side-stream
        mode, by Work.wait() on each of the two streams otherwise.  A third stream must wait for one of them itself."""
        self.wait()
        counts = frames_per_rank(self.n_frames, self.world)
        r, acc = 0, 0
        while frame_id >= acc + counts[r]:
            acc += counts[r]
            r += 1
        i = frame_id - acc
        n = int(self.sizes[r, i].item())
        return self.gathered[r * max(self.slots, 1) + i, :n]
