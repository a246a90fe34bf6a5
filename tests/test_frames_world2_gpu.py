"""N > 1 with REAL solvers on the one GPU a test box has: two ranks (`gloo` -- RCCL refuses two ranks per device), both on device 0,
each rank a real `Regularizer` on its own stream with its own 640x480 frame (bench.py's per-rank seeds), the solver leaving
x * graph_scale in the gather's send row from its own launch (flame_nltgv2_set_export_target; reference contract: one read-back per
frame, flame.cc:372-380) -> IdepthGather.gather(async) -> settle().  One rank's run is made to time out in one step
(FLAME_NLTGV2_OPT_FAULT_INJECT), so the re-gather path runs ACROSS ranks: the rank that replayed tells the other through the one-word
all-reduce, both gather again.  Every gathered row on every rank is compared bit for bit with the CPU checker.

Also the bench.py N = 2 code path, dry-run the way the driver launches it (torch.distributed.run --nproc-per-node 2) with the backend
override FLAME_BENCH_BACKEND=gloo and both ranks on device 0 (FLAME_BENCH_DEVICE=0): the JSON line an 8-GPU driver will parse --
value = world * steps * iters / wall, `result_gather`, no `cpu_baseline` -- is produced once before it meets real hardware."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from tests.conftest import ROOT

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q, backend="gloo", one_device=True, side_stream="1"):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["FLAME_GATHER_SIDE_STREAM"] = side_stream  # (read by IdepthGather: the collective from a side stream, or by torch's own events)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch
    import torch.distributed as dist

    if backend == "nccl":
        torch.cuda.set_device(rank)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", rank))
    else:
        dist.init_process_group(backend, rank=rank, world_size=world)
    try:
        import flame_amd
        from flame_amd import synth
        from flame_amd.frames import IdepthGather
        from flame_amd.regularizer import OPT_FAULT_INJECT
        from oracle import capi as oracle

        dev = torch.device("cuda", 0 if one_device else rank)  # (one_device: the box has one GPU, every rank uses it)
        torch.cuda.set_device(dev)
        iters = 200
        frames = [synth.make_graph("640x480", seed=1234 + r) for r in range(world)]  # bench.py's per-rank seeds
        refs = [synth.copy_graph(g) for g in frames]
        g = frames[rank]
        reg = flame_amd.Regularizer(dev.index)
        stream = torch.cuda.Stream(device=dev, priority=-1)
        reg.set_stream(stream.cuda_stream)
        reg.upload_graph(g)
        ig = IdepthGather(dist, [g["V"]], world, dev, stream=stream)
        p = flame_amd.Params()
        ok, regathers, paths = True, [], []
        for step in range(4):
            fault = step == 2 and rank == world - 1  # the last rank's run times out in step 2: the others must learn of it and gather again too
            if fault:
                reg.set_option(OPT_FAULT_INJECT, 3000)
            reg.set_export_target(ig.local_row(0).data_ptr(), 1.0)
            reg.run_async(p, iters)
            paths.append(reg.info()["last_run_path"])
            with torch.cuda.stream(stream):
                ig.gather(async_op=True, regs=[reg])
            for ref in refs:
                oracle.omp_run(ref, iters, min(8, os.cpu_count() or 1))  # (bit-identical to the sequential checker)
            regathers.append(ig.settle([reg]))
            if fault:
                reg.set_option(OPT_FAULT_INJECT, 0)
            for r, ref in enumerate(refs):
                got = ig.frame(r).cpu().numpy()
                ok = ok and got.shape[0] == ref["V"] and np.array_equal(got, ref["x"])
        state_ok = bool(np.array_equal(reg.download_state(("x",))["x"], refs[rank]["x"]))
        info = reg.info()
        reg.close()
        q.put((rank, ok, state_ok, regathers, paths, info["timeouts_recovered"], str(dist.get_backend())))
    finally:
        dist.destroy_process_group()


def test_world2_real_solvers_on_one_device_gather_and_regather_across_ranks():
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=600) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, ok, state_ok, regathers, paths, recovered, backend in res:
        assert backend == "gloo"
        assert ok, f"rank {rank}: a gathered row differs from the CPU checker"
        assert state_ok, f"rank {rank}: solver state differs from the CPU checker"
        assert regathers[2] == 1, (rank, regathers)       # both ranks gathered again after rank 1's replay
        assert paths[0] in (6, 7), (rank, paths)          # a persistent launch wrote the send row itself
    assert res[0][3] == res[1][3], "the ranks disagree on which steps were re-gathered"
    assert res[1][5] >= 1                                 # rank 1 (the last) did take a run back
    # (two processes share the GPU here: a run of either rank may also expire on its own -- it is then redone bit-identically and
    #  re-gathered, which the per-row comparison above covers; only the injected one is asserted by step)


@pytest.mark.parametrize("side_stream", ["1", "0"])
def test_one_rccl_rank_per_visible_device(side_stream):
    """(Both ways of issuing the collective: FLAME_GATHER_SIDE_STREAM = 1, the default, and 0.)  The same exchange over RCCL with one rank per GPU, on min(visible devices, 8) ranks -- skipped on a one-GPU box (RCCL refuses
    two ranks per device), so an 8-GPU node runs this path in the GPU test tier before the scaling bench meets it: every rank a real
    solver on its own device, export row -> all-gather on the solver's stream -> settle(), the last rank's run made to time out in one
    step (every rank gathers again), every gathered row on every rank against the CPU checker."""
    import torch
    import torch.multiprocessing as mp

    world = min(torch.cuda.device_count(), 8)
    if world < 2:
        pytest.skip("one visible device: RCCL wants one device per rank (the two-rank gloo test covers the logic)")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, "nccl", False, side_stream)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=900) for _ in procs)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    for rank, ok, state_ok, regathers, paths, recovered, backend in res:
        assert backend == "nccl"
        assert ok, f"rank {rank}: a gathered row differs from the CPU checker"
        assert state_ok, f"rank {rank}: solver state differs from the CPU checker"
        assert regathers[2] == 1, (rank, regathers)
        assert paths[0] in (6, 7), (rank, paths)
    assert all(r[3] == res[0][3] for r in res), "the ranks disagree on which steps were re-gathered"
    assert res[-1][5] >= 1


def test_bench_two_ranks_dry_run_with_backend_override():
    env = dict(os.environ)
    env.update({"FLAME_BENCH_BACKEND": "gloo", "FLAME_BENCH_DEVICE": "0", "HSA_ENABLE_IPC_MODE_LEGACY": "0"})
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "6", "--warmup", "2"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{") and '"metric"' in ln]
    assert len(lines) == 1, r.stdout[-2000:]                       # ONE JSON line, from rank 0
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 6 and out["warmup"] == 2 and out["scaling"] == "weak"
    assert out["metric"].startswith("NLTGV2 primal-dual iters/sec") and out["unit"] == "iters/s"
    assert out["config"]["parallelism"] == "frames x2"
    # value = whole-job aggregate: world * steps * iters / wall (max over ranks)
    iters = out["config"]["iters_per_step"]
    assert abs(out["value"] - 2 * 6 * iters / (out["ms_per_step"] * 6e-3)) / out["value"] < 1e-3
    rg = out["result_gather"]
    assert rg["ranks"] == 2 and rg["last_row_matches_state"] is True and rg["backend"] == "gloo"
    # the N > 1 line carries a roofline that can be checked rank by rank: every rank's device time of a step and its fraction
    rf = out["roofline"]
    assert len(rf["per_rank_step_us"]) == 2 and all(u > 0 for u in rf["per_rank_step_us"]) and len(rf["per_rank_frac"]) == 2
    assert abs(rf["per_rank_step_us"][0] - rf["avg_launch_us"] * rf["launches_per_step"]) / rf["per_rank_step_us"][0] < 1e-3
    # (two processes share ONE device here: each rank's step is longer than alone; what is checked is that the whole-job value is the
    #  sum of what the ranks really did in the max-over-ranks wall time)
    assert out["value"] <= 2 * iters / (min(rf["per_rank_step_us"]) * 1e-6) * 1.001
    assert "cpu_baseline" not in out                                # reported at N = 1 only
    assert out["parity"]["bit_identical"] and out["parity"]["timed_context_bit_identical"]
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "r06_bench_torchrun_2ranks_gloo_one_device.json"), "w") as f:
        f.write(lines[0] + "\n")


def _bench_line(cmd, env, timeout=900):
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, cwd=ROOT, env=env)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{") and '"metric"' in ln]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


def test_one_rccl_rank_under_torchrun_is_as_fast_as_the_plain_line():
    """What the result gather costs before a node shows it: bench.py under torch.distributed.run with ONE rank (RCCL world 1: a real
    collective kernel per step, the export row, the double-buffered rows, the side stream) against the plain N = 1 line of the same
    box, same steps -- whole-job value within 5 % (round 5 measured 1.00; a gather that lands on the solver's queue again, or waits
    for the runs around it, shows up here as 0.9 or less)."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    args = [os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "30", "--warmup", "4", "--no-extras", "--no-cpu-baseline"]
    plain = max(_bench_line([sys.executable] + args, env)["value"] for _ in range(2))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port())] + args
    outs = [_bench_line(cmd, env) for _ in range(2)]
    best = max(o["value"] for o in outs)
    assert outs[0]["result_gather"]["backend"] == "nccl" and outs[0]["result_gather"]["last_row_matches_state"] is True
    assert outs[0]["roofline"]["per_rank_step_us"][0] > 0
    assert best >= 0.95 * plain, (best, plain, outs[0]["result_gather"])
