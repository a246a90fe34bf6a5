#!/usr/bin/env python3
"""bench.py -- NLTGV2 primal-dual iterations/sec on a 640x480 Delaunay graph (BASELINE.json metric).

A "step" is one pass of the hot path over one batch of synthetic input: `--iters` (200) x
step(params, graph) on one 640x480-derived Delaunay graph per GPU (BASELINE.json configs[1]; with
--gpus N each rank owns an independent frame = configs[3], results gathered with RCCL).  Inputs are
resident in HBM before the timed region.  Prints ONE JSON line on rank 0.

  python bench.py                         # 1 GPU, defaults finish in about a minute
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
      --master-port P bench.py --gpus N --steps K --warmup W

Extra keys beyond the driver contract: "roofline" (dominant kernel vs the 8 TB/s HBM peak, from the
algorithmic byte count 64*V+40*E per step), "cpu_baseline" (the CPU checker in the reference's
node-based layout, 1 thread, timed here on the box's host cores), "parity" (this run's result vs the
checker), "batched" (aggregate rate with many frames resident on one GPU), "other_configs".
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# Before the HIP runtime starts: more in-order hardware queues than the default 4, so that the streams of this process (solver, the
# context's builder and rasteriser streams, RCCL's, torch's) do not share one.  Round 4 needed this for the result gather to overlap the
# solve (0.35 instead of 0.21 ms per step without it); round 5 found the cause -- the gather's waits were issued on torch's DEFAULT stream
# instead of the solver's (IdepthGather(stream=...) now orders them itself) -- and the step loop is the same with 8, 4 or 1 queue
# (tools/overlap_probe.py, profiles/r05_gather_overlap.txt).  Kept as the roomier setting; nothing depends on it any more.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

HBM_PEAK_GBPS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--iters", type=int, default=200, help="primal-dual iterations per step (BASELINE: 200)")
    ap.add_argument("--config", default="640x480")
    ap.add_argument("--batch", type=int, default=64, help="frames per GPU in the extra 'batched' measurement (0 = skip)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip batched / other-config measurements")
    ap.add_argument("--cpu-iters", type=int, default=20000)
    ap.add_argument("--persistent", type=int, default=1, help="0: one launch per step; 1: persistent single launch (form by size); 3 / 4 / 6: the vertex-per-lane / patch-per-wave / two-half-edges form by name")
    return ap.parse_args()


def measure(reg, params, iters, steps, warmup, sync, barrier, stream, before_step=None, after_step=None):
    """warmup, then EXACTLY `steps` steps bracketed by barrier+synchronize.  The steps are enqueued back to back
    (no host round trip between them; run_async); ONE HIP event pair on the solver's stream brackets the `steps` launches
    (a pair around every launch put two more packets between two kernels of the solver's in-order queue: +10 us per step,
    tools/overlap_probe.py -- 5 % of what it measured).  Returns (wall seconds, device ms between the two events)."""
    import torch

    for _ in range(warmup):
        if before_step:
            before_step()
        reg.run(params, iters)
        if after_step:
            after_step()
    barrier()
    sync()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record(stream)
    for _ in range(steps):
        if before_step:
            before_step()
        reg.run_async(params, iters)
        if after_step:
            after_step()
    e1.record(stream)
    reg.sync()  # also checks the solver's error word
    sync()
    barrier()
    t1 = time.perf_counter()
    return t1 - t0, e0.elapsed_time(e1)


def main():
    a = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if a.gpus != world:
        if world == 1 and a.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

    import numpy as np
    import torch  # first: one HIP runtime per process

    import flame_amd
    from flame_amd import synth

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (there is no CPU fallback for the product path)")
    # Dry-run aids (never set by the driver): FLAME_BENCH_BACKEND=gloo runs the N > 1 code path where RCCL cannot (two ranks on
    # ONE device: tests/test_frames_world2_gpu.py), FLAME_BENCH_DEVICE=<ordinal> puts every rank on that device.
    backend = os.environ.get("FLAME_BENCH_BACKEND", "nccl")
    if os.environ.get("FLAME_BENCH_DEVICE") is not None:
        local_rank = int(os.environ["FLAME_BENCH_DEVICE"])
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1 or "TORCHELASTIC_RUN_ID" in os.environ or os.environ.get("FLAME_BENCH_FORCE_DIST"):
        # launched by torch.distributed.run (also with a single rank: exercises the RCCL path)
        import torch.distributed as dist

        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend=backend)

    def barrier():
        if dist is not None:
            dist.barrier()

    def sync():
        torch.cuda.synchronize()

    params = flame_amd.Params()
    # one independent frame per rank (BASELINE configs[3]); seeds differ so the frames differ
    g = synth.make_graph(a.config, seed=1234 + rank)
    t_c0 = time.perf_counter()
    reg = flame_amd.Regularizer(local_rank)
    reg.set_option(flame_amd.regularizer.OPT_PERSISTENT, a.persistent)
    t_c1 = time.perf_counter()
    reg.upload_graph(g)
    reg.run(params, a.iters)  # (the first frame of the process: upload + the first solve, settled)
    t_c2 = time.perf_counter()
    global _FIRST_FRAME
    _FIRST_FRAME = first_frame = {"process_first_context_create_ms": round((t_c1 - t_c0) * 1e3, 3), "process_first_frame_ms": round((t_c2 - t_c1) * 1e3, 3)}
    reg.upload_graph(g)  # (the timed loop starts from the initial state)
    info = reg.info()

    # the solver runs on a torch stream: torch's HIP events can then bracket its launches, and (N > 1) export ->
    # all_gather is ordered on the device without a host round trip between them
    # (high priority: the solver's waves win the issue slots they share with the overlapped gather's kernel -- a lock-step network
    #  runs at the pace of its slowest member; worth 1-4 % under torch.distributed.run, nothing without a gather)
    solver_stream = torch.cuda.Stream(device=local_rank, priority=-1)
    reg.set_stream(solver_stream.cuda_stream)
    before_step = after_step = None
    if dist is not None:
        # result gather (configs[3]): x*graph_scale of every rank's frame to all ranks, one RCCL
        # all_gather per step; no collective on the solve path
        from flame_amd.frames import IdepthGather

        ig = IdepthGather(dist, [g["V"]], world, torch.device("cuda", local_rank), stream=solver_stream)
        ig.check_overlap(reg, params)  # (warns if the collective does not run beside the solver)
        reg.upload_graph(g)


        def before_step():  # the solver leaves x * graph_scale in the gather's (double-buffered) send row itself
            reg.set_export_target(ig.local_row(0).data_ptr(), 1.0)

        def after_step():
            # overlaps the next step's solve; completed before the buffer is reused.  (The gather reads the row of a run
            # nobody has checked yet: a consumer of the gathered block calls ig.settle([reg]) first, which re-gathers if the
            # run had to be taken back and redone -- done once below, after the timed region.)
            ig.gather(async_op=True, regs=[reg])

    wall, ev_ms = measure(reg, params, a.iters, a.steps, a.warmup, sync, barrier, solver_stream, before_step, after_step)
    gather_us = None
    gather_ok = None
    if dist is not None:
        regathered = ig.settle([reg])
        # the last gathered block against this rank's own state: row `rank` must be x * 1.0 of the timed context
        mine = ig.frame(rank).cpu().numpy()
        gather_ok = bool(np.array_equal(mine, reg.download_state(("x",))["x"]))
        # latency of the result gather on its own (configs[3]: "RCCL gather over xGMI"), outside the timed region
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            ig.local_row(0)
            ig.gather(async_op=False)
            torch.cuda.synchronize()
        gather_us = (time.perf_counter() - t0) / 20 * 1e6
    run_path = flame_amd.regularizer.RUN_PATHS.get(reg.info()["last_run_path"], "?")
    per_rank_launch_us = None
    if dist is not None:
        t = torch.tensor([wall], device="cuda" if backend == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        wall = float(t.item())
        # every rank's own launch time (HIP events on ITS solver stream around its K launches): the N > 1 line carries a roofline that
        # can be checked rank by rank, the way the N = 1 line is
        mine = torch.zeros(world, device="cuda" if backend == "nccl" else "cpu", dtype=torch.float64)
        mine[rank] = ev_ms * 1e3 / a.steps
        dist.all_reduce(mine, op=dist.ReduceOp.SUM)
        per_rank_launch_us = [round(float(v), 3) for v in mine.tolist()]
    total_iters = world * a.steps * a.iters
    value = total_iters / wall

    out = None
    if rank == 0:
        B_iter = info["algorithmic_bytes_per_iter"]
        # Dominant kernel.  Persistent paths: ONE launch per step covers all `iters` primal-dual iterations, so algorithmic
        # bytes per launch = iters * (64V + 40E) and the average launch duration is the time between two HIP events on the
        # solver's stream around the K back-to-back launches, divided by K (it includes the gap between two launches).  Per-step path: one k_fused_step launch per
        # iteration; the event time divided by the launches then includes the ~3.5 us dependent-launch gaps.
        kernel = {"persistent-pv": "k_persistent_pv", "persistent-pv2": "k_persistent_pv2", "persistent-tv": "k_persistent_tv"}.get(run_path, "k_fused_step")
        persistent = run_path.startswith("persistent")
        launches_per_step = 1 if persistent else a.iters
        launch_us = ev_ms * 1e3 / (a.steps * launches_per_step)
        bytes_per_launch = B_iter * (a.iters if persistent else 1)
        achieved = bytes_per_launch / (launch_us * 1e-6) / 1e9
        per_iter_us = ev_ms * 1e3 / (a.steps * a.iters)
        roofline = {
            # what binds ONE small frame is the dependency latency of a step (one cross-CU hand-off + the instructions
            # between a record arriving and the next leaving), not bandwidth: `frac` is still quoted against the HBM
            # peak, as SURVEY.md 8(d) defines it, so that it is comparable with the throughput regime below
            "bound": "latency", "peak_of": "hbm", "kernel": kernel,
            "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBPS, 4), "traffic": measured_traffic(a.config, run_path),
            "algorithmic_bytes_per_launch": bytes_per_launch, "avg_launch_us": round(launch_us, 3),
            "launches_per_step": launches_per_step, "per_iteration_us": round(per_iter_us, 3),
            **({"per_rank_step_us": per_rank_launch_us,
                "per_rank_frac": [round(bytes_per_launch * launches_per_step / (u * 1e-6) / 1e9 / HBM_PEAK_GBPS, 4) for u in per_rank_launch_us],
                "per_rank_note": "device time of one step (its launches, HIP events on the rank's solver stream) on every rank, and the fraction of the HBM "
                                 "peak that rank's kernel reaches; `avg_launch_us` / `frac` above are rank 0's"} if per_rank_launch_us else {}),
            "note": "single frame: dependency-latency bound -- the whole state (1.6 MB) is register/LDS resident, HBM traffic "
                    "is a fraction of the algorithmic bytes, and each iteration waits one cross-CU neighbour hand-off; "
                    "'step_cycles' is the in-kernel cycle account of the same kernel, 'batched' the throughput regime",
        }
        if run_path == "persistent-pv":
            try:  # where the records that cross XCDs were put (FLAME_NLTGV2_OPT_PLACEMENT) and what the calibration measured
                pi = reg.placement_info()
                roofline["record_placement"] = {
                    "state": pi["state"], "placed_records": pi["placed_records"],
                    "cross_xcd_handoff_us_by_page": {"best": round(pi["best_us"], 3), "mean": round(pi["mean_us"], 3),
                                                     "worst": round(pi["worst_us"], 3)},
                    "note": "one-way hand-off of a 16-byte record between two XCDs, by the 4 KB page it lives on (mean over the "
                            "56 ordered XCD pairs, all pairs exchanging at once); records read across XCDs are put on the best "
                            "page of their pair, 128-byte aligned per producing patch"}
            except Exception as e:  # noqa: BLE001
                roofline["record_placement"] = f"{type(e).__name__}: {e}"
            # (--no-extras leaves both out: they launch the dominant kernel with other lengths and graphs, and the rocprofv3 statistics of that
            #  command -- tools/profile.sh, profiles/r06_kernel_stats_bench.csv -- are to hold the headline's launches only)
            try:
                if not a.no_extras:
                    roofline["step_cycles"] = pv_step_cycles(flame_amd, g, params, a.iters, local_rank)
            except Exception as e:  # noqa: BLE001
                roofline["step_cycles"] = f"{type(e).__name__}: {e}"
            try:  # what `frac` has to be read against: the physical floor of one iteration of ONE small frame, measured by this run
                if not a.no_extras:
                    roofline["floor"] = latency_floor(flame_amd, synth, params, a.iters, local_rank, B_iter, roofline)
            except Exception as e:  # noqa: BLE001
                roofline["floor"] = f"{type(e).__name__}: {e}"
        out = {
            "metric": "NLTGV2 primal-dual iters/sec on 640x480 Delaunay graph; depth RMS vs CPU",
            "value": round(value, 1), "unit": "iters/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(wall * 1e3 / a.steps, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{a.config} Delaunay graph, V={g['V']} E={g['E']}, {a.iters} primal-dual iters per step, "
                                   f"one frame per GPU", "iters_per_step": a.iters, "V": g["V"], "E": g["E"],
                       "parallelism": f"frames x{world}" if world > 1 else "single frame"},
            "roofline": roofline,
            "device": {"name": info["device_name"], "arch": info["gcn_arch"], "cus": info["compute_units"]},
            "run_path": run_path,
        }
        if gather_us is not None:
            out["result_gather"] = {"overlap": getattr(ig, "overlap", None), "all_gather_us": round(gather_us, 1), "bytes_per_rank": int(g["V"]) * 4, "ranks": world, "backend": backend,
                                    "last_row_matches_state": gather_ok, "regathered_after_replay": int(regathered),
                                    "note": "blocking all_gather_into_tensor of x*graph_scale incl. host launch + sync; in the "
                                            "step loop it is asynchronous and overlaps the next solve"}
        # ---- parity of THIS run's input against the CPU checker (same seeded input, same iteration count)
        from oracle import capi as oracle

        chk = flame_amd.Regularizer(local_rank)
        chk.upload_graph(g)
        chk.run(params, a.iters)
        got = chk.download_state(("x", "w1", "w2"))
        chk.close()
        ref = synth.copy_graph(g)
        oracle.run(ref, a.iters)
        d = got["x"].astype(np.float64) - ref["x"].astype(np.float64)
        out["parity"] = {"depth_rms_vs_cpu": float(np.sqrt(np.mean(d * d))), "max_abs": float(np.abs(d).max()),
                         "bit_identical": bool(all(np.array_equal(got[k], ref[k]) for k in got)),
                         "iters": a.iters, "tolerance_rms": 1e-4}
        # ... and the TIMED context itself: `reg` has by now done (warmup + steps) x iters iterations from the uploaded input
        # on the solver's stream (the launches `value` was measured on); its whole state must be what the checker has after as many
        total_run = (a.warmup + a.steps) * a.iters
        state_keys = ("x", "w1", "w2", "x_bar", "w1_bar", "w2_bar", "q1", "q2", "q3")
        timed_state = reg.download_state(state_keys)
        ref_t = synth.copy_graph(g)
        t_chk = time.perf_counter()
        threads = min(16, os.cpu_count() or 1)
        if threads >= 4:   # the OpenMP two-phase form of the checker: bit-identical to the sequential one for any thread count
            oracle.omp_run(ref_t, total_run, threads)
        else:
            oracle.run(ref_t, total_run)
        dt = timed_state["x"].astype(np.float64) - ref_t["x"].astype(np.float64)
        out["parity"]["timed_context_bit_identical"] = bool(all(np.array_equal(timed_state[k], ref_t[k]) for k in state_keys))
        out["parity"]["timed_context"] = {"iters": total_run, "depth_rms_vs_cpu": float(np.sqrt(np.mean(dt * dt))),
                                          "arrays_compared": list(state_keys), "checker_s": round(time.perf_counter() - t_chk, 2),
                                          "timeouts_recovered": int(reg.info()["timeouts_recovered"])}
        # ---- CPU baseline on this box's host cores: reference-layout restatement, 1 thread (the
        # reference runs its solver on exactly one thread, flame.cc:99-112)
        if not a.no_cpu_baseline and world == 1:  # the CPU baseline is reported at N=1 only
            secs = oracle.reflayout_run_timed(synth.copy_graph(g), a.cpu_iters)
            flat = oracle.run_timed(synth.copy_graph(g), max(200, a.cpu_iters // 4))
            out["cpu_baseline"] = {
                "value": round(a.cpu_iters / secs, 1), "unit": "iters/s", "cores": 1, "kind": "port",
                "sample": f"{a.cpu_iters} iterations of the same {a.config} graph (V={g['V']} E={g['E']}), "
                          f"node-based (BGL-like) layout, 1 thread, {secs:.1f} s",
                "flat_soa_1thread_iters_per_s": round(max(200, a.cpu_iters // 4) / flat, 1),
                "host_cores": os.cpu_count(),
            }
            out["speedup_vs_cpu_baseline"] = round(value / world / out["cpu_baseline"]["value"], 1)
            # the strongest fair CPU number: OpenMP two-phase form over the host cores, bit-identical
            # to the sequential restatement; the reference itself runs the solver on one thread
            best = None
            for t in (8, 16, 32, 64, 128):
                if t > (os.cpu_count() or 1):
                    break
                s_omp = oracle.omp_run_timed(synth.copy_graph(g), 2000, t)
                if best is None or 2000 / s_omp > best[0]:
                    best = (2000 / s_omp, t)
            if best:
                out["cpu_baseline"]["openmp_all_cores"] = {
                    "value": round(best[0], 1), "unit": "iters/s", "threads": best[1],
                    "sample": "2000 iterations per thread count in (8,16,32,64,128), best reported; flat SoA arrays, "
                              "edge-parallel dual + vertex-gather primal",
                    "gpu_over_this": round(value / world / best[0], 1)}
    if not a.no_extras and world == 1:
        try:  # labelled side measurements: never allowed to take the contract line down with them
            extras(a, reg, params, out, flame_amd, synth, sync, info)
        except Exception as e:  # noqa: BLE001
            out["extras_error"] = f"{type(e).__name__}: {e}"
        try:
            out["open_run"] = open_run_block(reg, params, solver_stream)
        except Exception as e:  # noqa: BLE001
            out["open_run"] = f"{type(e).__name__}: {e}"
    reg.close()
    if rank == 0:
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def open_run_block(reg, params, stream):
    """flame_nltgv2_run_open on the bench's own graph: the open run's kernel instance against the plain one (4 000 iterations per launch, an open run
    left to reach its bound) and how long an open run takes to stop once a call needs the state."""
    import time as _t

    import numpy as _np
    import torch

    if not reg.run_open(params, 64):
        return {"applicable": False}
    reg.sync()
    per = {"plain": [], "open": []}
    for _ in range(5):
        for kind in ("plain", "open"):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            if kind == "plain":
                reg.run_async(params, 4000)
            else:
                reg.run_open(params, 4000)
            e1.record(stream)
            torch.cuda.synchronize()
            reg.sync()
            per[kind].append(e0.elapsed_time(e1) * 1e3 / 4000)
    wall, its = [], []
    for _ in range(12):
        before = reg.iterations()[0]
        reg.run_open(params, 1 << 18)
        _t.sleep(0.0005)
        t0 = _t.perf_counter()
        reg.sync()
        wall.append((_t.perf_counter() - t0) * 1e6)
        its.append(reg.iterations()[0] - before)
    return {"applicable": True, "plain_us_per_iteration": round(float(_np.median(per["plain"][1:])), 4),
            "open_us_per_iteration": round(float(_np.median(per["open"][1:])), 4),
            "stop_latency_us": round(float(_np.median(wall)), 1), "iterations_when_stopped_after_0.5_ms": [int(min(its)), int(max(its))],
            "note": "one launch that iterates until a call needs the solver settled (the reference's solver thread is while(true) step()): a patch reads "
                    "the host's request every 64 iterations and publishes the iteration all patches leave at (+128); stop_latency = sync() issued 0.5 ms "
                    "after run_open, wall time until it returns; profiles/r06_holds.txt section 8"}


def pv_step_cycles(flame_amd, g, params, iters, device):
    """In-kernel cycle account of k_persistent_pv (FLAME_NLTGV2_OPT_PROBE) on this run's graph: per patch and step the
    shader cycles spent waiting for the neighbours' records, and the cycles from their arrival to the next publish."""
    import numpy as np

    from flame_amd.regularizer import OPT_PERSISTENT, OPT_PROBE

    r = flame_amd.Regularizer(device)
    try:
        r.set_option(OPT_PERSISTENT, 4)
        r.set_option(OPT_PROBE, 1)
        r.upload_graph(g)
        r.run(params, iters)
        r.run(params, iters)
        p = r.read_probe().reshape(-1, iters, 8).astype(np.int64)[:, iters // 10:, :]
        p = p[p[:, 0, 5] != 0]  # (idle padding behind an XCD's instances never runs a step)
        info = r.info()
    finally:
        r.close()
    wait, comp = p[:, :, 2].mean(axis=1), p[:, :, 3].mean(axis=1)
    period = float((np.diff(p[0, :, 5]) & 0xffffffff).mean())
    us = float((np.diff(p[0, :, 6]) & 0xffffffff).mean()) / 100.0
    crit = int(np.argmin(wait))
    slow = int(np.argmax(comp))
    deg = p[:, 0, 7] & 0xff  # (probe word 7: the patch's largest vertex degree = the length of its ordered-sum chain | fetch lanes << 8)
    return {"period": round(period, 0), "period_us": round(us, 3), "shader_clock_GHz": round(period / (us * 1e3), 3),
            "compute_median": round(float(np.median(comp)), 0), "compute_max": round(float(comp.max()), 0),
            "compute_max_over_median": round(float(comp.max() / np.median(comp)), 3),
            "slowest_patch_largest_degree": int(deg[slow]), "median_patch_largest_degree": int(np.median(deg)),
            "compute_by_largest_degree": {str(int(d)): round(float(comp[deg == d].mean()), 0) for d in sorted(set(deg.tolist()))},
            "wait_median": round(float(np.median(wait)), 0), "wait_min": round(float(wait.min()), 0),
            "least_slack_patch": {"compute": round(float(comp[crit]), 0), "wait": round(float(wait[crit]), 0)},
            "poll_rounds_per_step": round(float(p[:, :, 4].mean()), 2), "patches": int(info["he_waves"]),
            "instances": int(p.shape[0]),
            "note": "cycles per step with the probe compiled in (+7-9 %: floor.probe_overhead_us); the lock-step network runs at the pace of its least-slack patches: period = their "
                    "compute + their wait (one hand-off).  A patch's compute is the chain of its largest vertex: 2 x degree dependent additions in the "
                    "reference's edge order (`compute_by_largest_degree`) -- the slowest patch is the one that holds the graph's largest vertex, which no "
                    "packing can shorten; polling faster or slower than one load per LDS round trip lengthens the wait (profiles/r04_poll_variants.txt)"}


def latency_floor(flame_amd, synth, params, iters, device, B_iter, roofline):
    """The floor under one iteration of one small frame, measured by this run.  An iteration is one dependent exchange of
    16-byte records between neighbouring patches (intrinsic: x_bar of step t feeds step t + 1) plus the instructions between
    a record arriving and the next one leaving; the exchange crosses an XCD border wherever the graph does.
      * cross-XCD hand-off: what this context's page calibration measured (k_place_calibrate, all 56 XCD pairs at once);
      * uncoupled period: eight DISJOINT graphs of the same total size, one per XCD -- no record crosses an XCD -- run by the
        same kernel: the period the coupled graph would have without its crossings; its least-slack patch's wait is the
        same-XCD hand-off as the kernel sees it (publish -> arrival, detection included);
      * HBM period: the algorithmic bytes of one iteration at the HBM peak -- what `frac` divides by."""
    import numpy as np

    from flame_amd.regularizer import OPT_PERSISTENT, OPT_PROBE

    def small(w, h, seed):
        pos = synth.make_points(w, h, 6, seed)
        return synth.assemble_graph(pos, synth.make_data_term(pos, w, h, seed), synth.delaunay_edges_native(pos))

    g8 = synth.concat_graphs([small(228, 168, 100 + k) for k in range(8)])  # 8 x ~1060 vertices ~ one 640x480 frame
    r = flame_amd.Regularizer(device)
    try:
        r.set_option(OPT_PERSISTENT, 4)
        r.upload_graph(g8)
        r.run(params, iters)
        ms, _ = timed_launches(r, params, iters, 5)
        ms_long, _ = timed_launches(r, params, 10 * iters, 3)  # (ten times the iterations per launch: the launch's fixed part amortised)
        path = flame_amd.regularizer.RUN_PATHS.get(r.info()["last_run_path"], "?")
    finally:
        r.close()
    r = flame_amd.Regularizer(device)
    try:
        r.set_option(OPT_PERSISTENT, 4)
        r.set_option(OPT_PROBE, 1)
        r.upload_graph(g8)
        r.run(params, iters)
        r.run(params, iters)
        p = r.read_probe().reshape(-1, iters, 8).astype(np.int64)[:, iters // 10:, :]
        p = p[p[:, 0, 5] != 0]
    finally:
        r.close()
    wait = p[:, :, 2].mean(axis=1)
    ghz = float((np.diff(p[0, :, 5]) & 0xffffffff).mean()) / (float((np.diff(p[0, :, 6]) & 0xffffffff).mean()) * 10.0)
    uncoupled_us = ms * 1e3 / iters
    probed_us = float((np.diff(p[0, :, 6]) & 0xffffffff).mean()) / 100.0  # the probed instance's own period in steady state, by the 100 MHz clock
    steady_us = ms_long * 1e3 / (10 * iters)                             # the plain instance's, with the launch's fixed part amortised
    probe_overhead_us = max(0.0, probed_us - steady_us)
    out = {"uncoupled_period_us": round(uncoupled_us, 3), "uncoupled_run_path": path,
           "uncoupled_graphs": f"8 disjoint Delaunay graphs, V={g8['V']} E={g8['E']} in total, one per XCD",
           "same_xcd_handoff_us": round(float(wait.min()) / (ghz * 1e3), 3),
           "uncoupled_steady_period_us": round(steady_us, 3), "probed_period_us": round(probed_us, 3), "probe_overhead_us": round(probe_overhead_us, 3),
           "same_xcd_handoff_less_probe_us": round(max(0.0, float(wait.min()) / (ghz * 1e3) - probe_overhead_us), 3),
           "hbm_period_us": round(B_iter / (HBM_PEAK_GBPS * 1e9) * 1e6, 3),
           "measured_period_us": roofline["per_iteration_us"]}
    rp = roofline.get("record_placement")
    if isinstance(rp, dict):
        out["cross_xcd_handoff_us"] = rp["cross_xcd_handoff_us_by_page"]
    sc = roofline.get("step_cycles")
    if isinstance(sc, dict):  # what the coupled frame's least-slack wait holds beyond a same-XCD hand-off: the crossing + later detection
        out["wait_min_minus_same_xcd_handoff_cycles"] = round(sc["wait_min"] - float(wait.min()), 0)
    out["frac_of_hbm_at_uncoupled_period"] = round(out["hbm_period_us"] / uncoupled_us, 4)
    out["note"] = ("one small frame cannot iterate faster than one hand-off plus the ~600 cycles of dependent instructions behind it; "
                   "`frac_of_hbm_at_uncoupled_period` is the roofline fraction this frame would show if none of its records crossed an XCD.  "
                   "`same_xcd_handoff_us` is the least-slack patch's wait as the PROBED instance sees it; the probe's own stamps and log store sit "
                   "inside that wait (`probe_overhead_us` = its steady period minus the plain instance's at ten times the iterations per launch, round 6: "
                   "profiles/r06_handoff.txt section 5), "
                   "`same_xcd_handoff_less_probe_us` takes them out; tools/hop_bench hands ONE record over in 0.26-0.29 us, the lock-step network of "
                   "tools/net_bench.hip (19 records of 7.6 producers, the same poll statement and stores) in 0.32 us")
    return out


def measured_traffic(config, run_path):
    """HBM bytes per launch of the dominant kernel from rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE),
    measured offline with the same command and committed under profiles/ (the counters cannot be read
    from inside the process).  None when no measurement is committed for this workload/path."""
    path = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        with open(path) as f:
            t = json.load(f)
        return t.get(f"{config}:{run_path}", {}).get("hbm_bytes_per_launch")
    except Exception:
        return None


VALU_PEAK_WINST_PER_S = 256 * 4 * 1.2e9  # 256 CUs x 4 SIMD-32, one wave64 VALU instruction per 2 cycles at 2.4 GHz
#                                          (MI355X_MICROARCH.md "Wave scheduling"; x 64 lanes x 2 flop = the 157.3 TFLOP/s fp32 peak)


def measured_counters(key):
    """The committed rocprofv3 PMC record of a workload (profiles/traffic.json: FETCH_SIZE/WRITE_SIZE bytes and SQ_INSTS_VALU
    per launch of its dominant kernel), or {}."""
    try:
        with open(os.path.join(ROOT, "profiles", "traffic.json")) as f:
            return json.load(f).get(key, {})
    except Exception:
        return {}


def valu_roofline(key, launch_s):
    """VALU-issue roofline of one launch: SQ_INSTS_VALU (wave instructions, measured offline with the same command) over
    the launch time measured now, against the chip's VALU issue rate."""
    n = measured_counters(key).get("valu_insts_per_launch")
    if not n or launch_s <= 0:
        return None
    rate = n / launch_s
    return {"bound": "valu", "insts_per_launch": int(n), "achieved": round(rate / 1e9, 1), "peak": round(VALU_PEAK_WINST_PER_S / 1e9, 1),
            "unit": "G wave-instr/s", "frac": round(rate / VALU_PEAK_WINST_PER_S, 4), "source": "profiles/traffic.json"}


def timed_launches(r, params, iters, n=10):
    """(mean, min) duration in ms of n launches of `iters` iterations, each bracketed by HIP events on the solver's stream.
    Fractions are computed from the MEAN, so that they agree with the rocprofv3 average of the same command under profiles/."""
    ts = [r.run_timed(params, iters) for _ in range(n)]
    return sum(ts) / len(ts), min(ts)


def reg_profile_kernel(reg, params, n=400):
    """Mean duration (us) of ONE k_fused_step launch: eager launches, hipGraph off, each `run(1)`
    bracketed by HIP events on the solver's stream (flame_nltgv2_run_timed)."""
    from flame_amd.regularizer import OPT_USE_HIPGRAPH

    reg.set_option(OPT_USE_HIPGRAPH, 0)
    for _ in range(20):
        reg.run_timed(params, 1)
    ts = sorted(reg.run_timed(params, 1) for _ in range(n))
    reg.set_option(OPT_USE_HIPGRAPH, 1)
    core = ts[n // 10: n - n // 10]  # trimmed mean: drops the occasional host hiccup
    return 1e3 * sum(core) / len(core)


_FIRST_FRAME = {}


def onchip_roofline(valu, oc, waves_per_cu):
    """What binds a register / LDS resident kernel, from the two VALU counters that disagree by a factor (verdict r4 #7):

    * `valu_pipe_throughput_frac` = SQ_INSTS_VALU / time / (1024 SIMD-32 x 2.4 GHz / 2): the share of the vector pipes' ISSUE SLOTS
      used (a wave64 instruction takes a SIMD-32 two cycles).  A third at 30 frames: the pipes are not the limit.
    * `wave_issue_busy_frac` = SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES x waves per SIMD: the share of SIMD cycles in which one of its
      waves has a vector instruction IN FLIGHT, issue to completion.  A lone wave issues one instruction per ~5 cycles (round 5's
      diagnostics of the region form, profiles/r05_wg_region.txt: 5 cycles x instructions + LDS latencies reproduces the measured step), because each
      instruction of its chain waits for the one before; W waves per SIMD interleave W such chains.  When this fraction nears 1 every
      cycle of the SIMD is covered by some wave's dependent instruction: adding waves no longer adds progress, although half the issue
      slots stay empty.  THIS is the binding measure of these kernels (wave64 on SIMD-32 with dependent chains), and `frac`."""
    a = (oc or {}).get("active_inst_valu_over_wave_cycles")
    wps = waves_per_cu / 4.0 if waves_per_cu else None
    busy = round(min(1.0, a * wps), 4) if a is not None and wps else None
    return {"bound": "wave-issue", "frac": busy, "wave_issue_busy_frac": busy, "waves_per_simd": wps,
            "valu_pipe_throughput_frac": valu["frac"] if valu else None,
            "note": "the state is register / LDS resident: the HBM figure above counts algorithmic bytes and may pass 1.  frac = the share of "
                    "SIMD cycles covered by a vector instruction in flight of one of its waves (dependent chains: ~5 cycles per instruction and "
                    "wave); the share of the vector pipes' issue slots used is valu_pipe_throughput_frac -- DESIGN.md section 6"}


def extras(a, reg, params, out, flame_amd, synth, sync, info):
    """Rank-0-only extra measurements (not part of `value`)."""
    # (1) batched mode: B independent frames resident on one GPU as a disjoint union.
    #     "resident": as many frames as the vertex-per-lane persistent kernel keeps in registers
    #     "large": a.batch frames -- more than fit at once: run as groups of resident frames, one
    #     persistent launch per group (frames are independent, so this is the same computation)
    out["batched"] = {}
    #     "ten": ten frames in ONE launch of the patch-per-wave kernel with two half-edges per lane (18.6 waves per CU)
    for label, nf, iters in (("ten", 10, 200), ("resident", None, 200), ("large", a.batch, 100)):
        if label == "resident":  # as many frames as the register-resident persistent kernel holds
            nf = max(1, info["tv_wave_capacity"] // max(1, info["tv_waves"] + 1))
        if not nf:
            continue
        frames = [synth.make_graph(a.config, seed=5000 + i) for i in range(nf)]
        union = synth.concat_graphs(frames)
        b = flame_amd.Regularizer(0)
        b.upload_graph(union)
        bi = b.info()
        b.run(params, iters)
        ms, ms_min = timed_launches(b, params, iters)
        path = flame_amd.regularizer.RUN_PATHS.get(b.info()["last_run_path"], "?")
        per_iter_us = ms * 1e3 / iters
        gbps = bi["algorithmic_bytes_per_iter"] / (per_iter_us * 1e-6) / 1e9
        out["batched"][label] = {
            "frames": nf, "V": bi["V"], "E": bi["E"], "run_path": path, "launch_groups": b.info()["last_run_groups"],
            "frame_iters_per_s": round(nf * iters / (ms * 1e-3), 1), "per_iteration_us": round(per_iter_us, 2),
            "achieved_GBps": round(gbps, 1), "frac": round(gbps / HBM_PEAK_GBPS, 4), "bound": "on-chip", "peak_of": "hbm",
            "timing": "mean of 10 launches (HIP events)", "best_launch_per_iteration_us": round(ms_min * 1e3 / iters, 2),
            "note": "frac counts ALGORITHMIC bytes (64 V + 40 E per iteration) against the HBM peak; the state is register / LDS "
                    "resident, the real HBM traffic is `traffic` (a tenth of it), so the rate may pass the HBM peak: what binds "
                    "is instruction issue (`valu`) and the neighbour exchange, not memory",
        }
        groups = max(1, b.info()["last_run_groups"])
        key = f"{a.config}x{nf}:{path}"
        out["batched"][label]["traffic"] = measured_counters(key).get("hbm_bytes_per_launch")
        out["batched"][label]["algorithmic_bytes_per_launch"] = int(bi["algorithmic_bytes_per_iter"] * iters / groups)
        out["batched"][label]["valu"] = valu_roofline(key, ms * 1e-3 / groups)
        oc = measured_counters(key).get("on_chip")
        if oc:  # rocprofv3 SQ counters of the same command (profiles/r05_counters.json): which on-chip resource the resident kernel keeps busy
            out["batched"][label]["on_chip"] = dict(oc, source="profiles/traffic.json", note="fractions of wave-cycles with an instruction of the class in flight "
                                                    "(LDS / VALU / VMEM / scalar), LDS bank-conflict share, LDS pipe busy fraction of the chip")
            v = out["batched"][label]["valu"]
            out["batched"][label]["roofline"] = onchip_roofline(v, oc, b.info()["last_run_waves_per_cu"])
        b.close()
    # (2) the other single-GPU BASELINE configs, 200 iterations each
    oc = {}
    for cfg in ("1280x720", "1920x1080"):
        g = synth.make_graph(cfg, seed=1234)
        r = flame_amd.Regularizer(0)
        r.upload_graph(g)
        fused = cfg == "1920x1080"
        if fused:  # BASELINE config 5: the photometric residual of the final x comes out of the solver's own launch
            import numpy as _np2

            from flame_amd import synth_stereo as _ss

            wv, hv, _ = synth.CONFIGS[cfg]
            tex = _ss.texture(wv, hv, 77, margin=0)
            ref_img = _np2.clip(_np2.rint(tex), 0, 255).astype(_np2.uint8)
            cmp_img = _np2.roll(ref_img, 3, axis=1)
            Kc = _np2.array([[0.52 * wv, 0, wv / 2.0], [0, 0.52 * wv, hv / 2.0], [0, 0, 1]])
            Kt = (Kc @ _np2.array([0.04, -0.01, 0.003])).astype(_np2.float32)
            r.photo_set_images(ref_img, cmp_img)
            r.photo_fuse(_np2.eye(3, dtype=_np2.float32), Kt, graph_scale=1.0, border=4)
        r.run(params, 200)
        ms, ms_min = timed_launches(r, params, 200)
        bi = r.info()
        gbps = bi["algorithmic_bytes_per_iter"] * 200 / (ms * 1e-3) / 1e9
        oc[cfg] = {"V": g["V"], "E": g["E"], "iters_per_s": round(200 / (ms * 1e-3), 1),
                   "run_path": flame_amd.regularizer.RUN_PATHS.get(bi["last_run_path"], "?"),
                   "achieved_GBps": round(gbps, 1), "frac": round(gbps / HBM_PEAK_GBPS, 4)}
        key = f"{cfg}:{oc[cfg]['run_path']}"
        oc[cfg]["bound"] = "latency"  # (one frame: the state is on chip, each iteration waits one neighbour hand-off)
        oc[cfg]["peak_of"] = "hbm"
        oc[cfg]["traffic"] = measured_counters(key).get("hbm_bytes_per_launch")
        oc[cfg]["algorithmic_bytes_per_launch"] = int(bi["algorithmic_bytes_per_iter"] * 200)
        oc[cfg]["avg_launch_us"] = round(ms * 1e3, 1)
        oc[cfg]["best_launch_us"] = round(ms_min * 1e3, 1)
        oc[cfg]["timing"] = "mean of 10 launches (HIP events)"
        oc[cfg]["valu"] = valu_roofline(key, ms * 1e-3)
        ocx = measured_counters(key).get("on_chip")
        if ocx:
            oc[cfg]["on_chip_roofline"] = onchip_roofline(oc[cfg]["valu"], ocx, bi["last_run_waves_per_cu"])
        if fused:
            res = r.photo_residual_last()
            oc[cfg]["photometric_residual"] = {"fused_into_the_solver_launch": True,
                                               "defined_on_fraction_of_vertices": round(float(_np2.isfinite(res).mean()), 3)}
        r.close()
    try:  # 1280x720 against its own floor: eight DISJOINT graphs of the same total size, one per XCD (no record crosses an XCD), same kernel
        def small(w, h, seed):
            pos = synth.make_points(w, h, 6, seed)
            return synth.assemble_graph(pos, synth.make_data_term(pos, w, h, seed), synth.delaunay_edges_native(pos))

        g8 = synth.concat_graphs([small(334, 250, 300 + k) for k in range(8)])
        r = flame_amd.Regularizer(0)
        r.set_option(flame_amd.regularizer.OPT_PERSISTENT, 4)
        r.upload_graph(g8)
        r.run(params, 200)
        ms8, _ = timed_launches(r, params, 200, 5)
        i8 = r.info()
        r.close()
        B720 = oc["1280x720"]["algorithmic_bytes_per_launch"] / 200
        oc["1280x720"]["floor"] = {
            "uncoupled_period_us": round(ms8 * 1e3 / 200, 3), "uncoupled_graphs": f"8 disjoint Delaunay graphs, V={g8['V']} E={g8['E']} in total, one per XCD",
            "uncoupled_run_path": flame_amd.regularizer.RUN_PATHS.get(i8["last_run_path"], "?"),
            "measured_period_us": round(1e6 / oc["1280x720"]["iters_per_s"], 3),
            "frac_of_hbm_at_uncoupled_period": round(B720 / (ms8 * 1e-3 / 200) / 1e9 / HBM_PEAK_GBPS, 4),
            "note": "what this frame would show if none of its records crossed an XCD: the period of a graph of this size is set by the patches a CU "
                    "holds (8 per CU here: two waves per SIMD), not by bandwidth and not by the crossings (DESIGN.md section 4, 'What the period depends on'; docs/DESIGN_r3.md for the measurements)"}
    except Exception as e:  # the extras never take the line down
        oc["1280x720"]["floor"] = f"{type(e).__name__}: {e}"
    out["other_configs"] = oc
    # (3) what the boundary costs when the host hands over fresh buffers every frame (PCIe-inclusive;
    #     never part of `value`): upload = host pack + H2D + device pack, download = unpack + D2H
    import time as _t

    g = synth.make_graph(a.config, seed=4321)
    first_frame = dict(_FIRST_FRAME)
    t_f0 = _t.perf_counter()
    r = flame_amd.Regularizer(0)
    t_f1 = _t.perf_counter()
    r.upload_graph(g)
    r.run(params, a.iters)
    t_f2 = _t.perf_counter()
    r.upload_graph(g)  # what a graph reset costs a running context (the reference resets whenever fewer than 3 features survive, flame.cc:281-290)
    r.run(params, a.iters)
    t_f3 = _t.perf_counter()
    first_frame.update({"fresh_context_create_ms": round((t_f1 - t_f0) * 1e3, 3), "first_frame_ms": round((t_f2 - t_f1) * 1e3, 3),
                        "graph_reset_frame_ms": round((t_f3 - t_f2) * 1e3, 3),
                        "note": "upload_graph + 200 iterations, settled: the first frame of the process (code objects, cooperative-launch set-up and the "
                                "record-placement calibration are taken by flame_nltgv2_create), of a fresh context in a running process, and of a "
                                "running context whose graph is reset"})
    up_call = []
    for _ in range(10):
        tc = _t.perf_counter()
        r.upload_graph(g)  # returns once everything is staged and enqueued ...
        up_call.append((_t.perf_counter() - tc) * 1e3)
        r.sync()
    t0 = _t.perf_counter()
    for _ in range(10):
        r.upload_graph(g)
        r.sync()           # ... this is until the device has it all
    t1 = _t.perf_counter()
    for _ in range(10):
        r.run(params, 1)
        r.download_state()
    t2 = _t.perf_counter()
    for _ in range(10):
        r.run(params, 1)
    t3 = _t.perf_counter()
    up_ms, down_ms = (t1 - t0) * 100, ((t2 - t1) - (t3 - t2)) * 100
    up_call_ms = sorted(up_call)[len(up_call) // 2]
    r.run(params, a.iters)
    ms = min(r.run_timed(params, a.iters) for _ in range(5))
    # per-frame warm-start synchronisation (syncGraph's graph edits): ~8 % vertex churn, new triangulation
    import numpy as _np

    rng = _np.random.default_rng(3)
    keep = rng.random(g["V"]) > 0.08
    fid = _np.nonzero(keep)[0].astype(_np.int32)
    pos2 = (g["pos"][keep] + rng.normal(0, 0.3, (len(fid), 2))).astype(_np.float32)
    n_new = g["V"] - len(fid)
    w_, h_, _c = synth.CONFIGS[a.config]
    pos2 = _np.concatenate([pos2, _np.stack([rng.random(n_new) * (w_ - 8) + 4, rng.random(n_new) * (h_ - 8) + 4], 1)
                            .astype(_np.float32)])
    fid = _np.concatenate([fid, _np.arange(g["V"], g["V"] + n_new, dtype=_np.int32)])
    data2 = _np.concatenate([g["data_term"][keep], _np.ones(n_new, _np.float32)])
    t_tri = []
    tri_buf = (_np.empty((2 * len(pos2), 3), _np.int32), _np.empty((3 * len(pos2), 2), _np.int32))  # (a frame loop keeps its output arrays)
    for _ in range(61):  # (the library's default: merged strips; the synthetic graphs pin their own order: synth.delaunay_native)
        t_d0 = _t.perf_counter()
        tris2, edges2 = flame_amd.delaunay(pos2, out=tri_buf)
        t_tri.append((_t.perf_counter() - t_d0) * 1e3)
    t_sorted = sorted(t_tri[1:])
    # ... and as a frame loop meets it: the worker pool asleep for a frame between two calls
    t_paced = []
    for _ in range(40):
        _t.sleep(0.002)
        t_d0 = _t.perf_counter()
        flame_amd.delaunay(pos2, out=tri_buf)
        t_paced.append((_t.perf_counter() - t_d0) * 1e3)
    t_paced.sort()
    out["delaunay"] = {"triangulate_ms": round(t_sorted[len(t_sorted) // 2], 3), "p95_ms": round(t_sorted[int(0.95 * (len(t_sorted) - 1))], 3),
                       "max_ms": round(t_sorted[-1], 3), "calls": len(t_sorted), "first_call_ms": round(t_tri[0], 3),
                       "every_2_ms": {"median_ms": round(t_paced[len(t_paced) // 2], 3), "p95_ms": round(t_paced[int(0.95 * (len(t_paced) - 1))], 3), "calls": len(t_paced)},
                       "points": int(len(pos2)), "triangles": int(len(tris2)),
                       "note": "host code (the reference's Triangle is host code too), exact predicates; median / p95 of back-to-back calls, and of calls "
                               "2 ms apart (the worker pool has gone to sleep in between, as in a frame loop)"}
    r.upload_graph(g)
    r.run(params, 50)
    fid0, edges0 = _np.arange(g["V"], dtype=_np.int32), _np.stack([g["src"], g["dst"]], 1)
    ones2 = _np.ones(len(fid), _np.float32)
    r.sync_graph(fid0, g["pos"], g["data_term"], g["data_weight"], edges0)  # warm-up of the sync path (allocations)
    sync_call, sync_done = [], []
    for _ in range(5):  # frame A -> frame B (8 % churn) timed, B -> A back untimed
        r.run(params, 50)
        t4 = _t.perf_counter()
        r.sync_graph(fid, pos2, data2, ones2, edges2, edges_unique=True)  # (the triangulator's own edge list)
        t5 = _t.perf_counter()
        r.sync()
        sync_call.append((t5 - t4) * 1e3)
        sync_done.append((_t.perf_counter() - t4) * 1e3)
        timed_path = r.info()["last_sync_path"]
        r.run(params, 50)
        r.sync_graph(fid0, g["pos"], g["data_term"], g["data_weight"], edges0)
    r.run(params, 50)
    r.sync_graph(fid, pos2, data2, ones2, edges2, edges_unique=True)
    r.run(params, 50)
    path_name = {1: "host", 2: "device"}
    out["frame_sync"] = {"sync_graph_ms": round(sorted(sync_done)[2], 3), "sync_graph_call_ms": round(sorted(sync_call)[2], 3),
                         "path": path_name.get(timed_path, "?"),
                         "churn": "8 % of vertices replaced, re-triangulated", "V": int(len(fid)), "E": int(r.info()["E"]),
                         "note": "median of 5; sync_graph_ms = until the device holds the new frame (call + stream sync), call_ms = until the call returns.  "
                                 "Round 4: index maps AND the new graph's layout tables are built by kernels over the resident previous topology "
                                 "(flame_amd/csrc/nltgv2_topo.hip); `host_path` = the same sync with both on the host (rounds 1-3); `two_halves` = "
                                 "sync_prepare (builder on a side stream, returns at once) / sync_commit (the solver stands still for this call only)"}
    # the same sync the host way, and in two halves with the solver iterating in between
    from flame_amd.regularizer import OPT_SYNC_PATH

    def timed_sync(prep_commit):
        calls = []
        for _ in range(5):
            r.run(params, 50)
            if prep_commit:
                t4 = _t.perf_counter()
                r.sync_prepare(fid, pos2, data2, ones2, edges2, edges_unique=True)
                t5 = _t.perf_counter()
                r.run_async(params, 400)   # the solver keeps iterating on the old graph
                _t.sleep(0.0005)           # (the builder runs on its side stream meanwhile)
                t6 = _t.perf_counter()
                r.sync_commit()
                calls.append(((t5 - t4) * 1e3, (_t.perf_counter() - t6) * 1e3))
            else:
                t4 = _t.perf_counter()
                r.sync_graph(fid, pos2, data2, ones2, edges2, edges_unique=True)
                r.sync()
                calls.append(((_t.perf_counter() - t4) * 1e3, 0.0))
            r.run(params, 50)
            r.sync_graph(fid0, g["pos"], g["data_term"], g["data_weight"], edges0)
        calls.sort()
        return calls[2]
    r.set_option(OPT_SYNC_PATH, 1)
    out["frame_sync"]["host_path"] = {"sync_graph_ms": round(timed_sync(False)[0], 3)}
    r.set_option(OPT_SYNC_PATH, 0)
    pc = timed_sync(True)
    out["frame_sync"]["two_halves"] = {"sync_prepare_call_ms": round(pc[0], 3), "sync_commit_call_ms": round(pc[1], 3),
                                       "note": "commit includes settling the 400 iterations enqueued between the halves"}
    # mesh -> dense idepthmap (utils::interpolateMesh, next row 8(f)-2), incl. the D2H copy of the map
    tris = tris2
    r.sync_graph(fid, pos2, data2, ones2, edges2, edges_unique=True)  # (the triangles belong to frame B's positions)
    r.run(params, 50)
    r.interpolate_mesh(tris, h_, w_)
    t6 = _t.perf_counter()
    for _ in range(10):
        img, cov = r.interpolate_mesh(tris, h_, w_)
    t7 = _t.perf_counter()
    out["rasterize"] = {"interpolate_mesh_ms": round((t7 - t6) * 100, 3), "triangles": int(len(tris)),
                        "image": f"{w_}x{h_}", "coverage": round(cov / (w_ * h_), 4)}
    if not a.no_cpu_baseline:
        from oracle import capi as _oracle

        xs = r.download_state(("x",))["x"]
        t8 = _t.perf_counter()
        _oracle.raster_interpolate_mesh(tris, pos2, xs, h_, w_)
        out["rasterize"]["cpu_checker_ms"] = round((_t.perf_counter() - t8) * 1e3, 3)
    out["host_boundary"] = {**first_frame, "upload_graph_ms": round(up_ms, 3), "upload_graph_call_ms": round(up_call_ms, 3), "download_state_ms": round(down_ms, 3),
                            "pcie_inclusive_iters_per_s": round(a.iters / ((ms + up_ms + down_ms) * 1e-3), 1),
                            "note": "upload+200 iters+download per frame; never reported as value"}
    r.close()
    out["feature_update"] = feature_update(a, w_, h_)


def feature_update(a, w_, h_):
    """Per-feature epipolar inverse-depth update (updateFeatureIDepths, next row 8(f)-4) on a synthetic plane scene:
    kernel time by HIP events, the host-array call (H2D + kernel + D2H of 40-byte records), Frame::create on the device."""
    import time as _t

    from flame_amd import synth_stereo as ss
    from flame_amd.stereo import FEATURE_DTYPE, FeatureTracker, StereoParams

    sc = ss.standard_scene(w_, h_)
    imgs = {c: sc.render(c) for c in (10, 11, 12)}
    feats = ss.make_features(sc, FEATURE_DTYPE, [10, 11], (w_ // 6) * (h_ // 6) // 2, 3)
    poses = ss.poses_for(sc, [10, 11], 12, 11)
    P = StereoParams()
    with FeatureTracker(sc.K32, sc.Kinv32, w_, h_) as tr:
        for c, img in imgs.items():
            tr.add_frame(c, img)
        t0 = _t.perf_counter()
        tr.add_frame(12, imgs[12])
        t_frame = _t.perf_counter() - t0
        k_ms, c_s, r_s, k1_ms, k16_ms = 1e9, 1e9, 1e9, 1e9, 1e9
        for _ in range(10):  # the default way in: the features stay on the device, the call returns the counters
            tr.set_features(feats)
            t0 = _t.perf_counter()
            _, st = tr.update_resident(P, 12, 11, poses)
            r_s = min(r_s, _t.perf_counter() - t0)
            k_ms = min(k_ms, tr.last_kernel_ms())
        for _ in range(5):   # a host array in and out (H2D + kernel + D2H of 40-byte records)
            f = feats.copy()
            t0 = _t.perf_counter()
            tr.update_feature_idepths(P, 12, 11, poses, f)
            c_s = min(c_s, _t.perf_counter() - t0)
        for lanes in (1, 16):
            tr.set_lanes_per_feature(lanes)
            for _ in range(5):
                tr.set_features(feats)
                tr.update_resident(P, 12, 11, poses)
                if lanes == 1:
                    k1_ms = min(k1_ms, tr.last_kernel_ms())
                else:
                    k16_ms = min(k16_ms, tr.last_kernel_ms())
    n = int(feats.shape[0])
    res = {"features": n, "image": f"{w_}x{h_}", "updated": st["num_idepth_updates"], "kernel_us": round(k_ms * 1e3, 2),
           "kernel_us_by_lanes_per_feature": {"1": round(k1_ms * 1e3, 2), "16": round(k16_ms * 1e3, 2)},
           "resident_call_us": round(r_s * 1e6, 1), "host_array_call_us": round(c_s * 1e6, 1),
           "frame_create_us": round(t_frame * 1e6, 1), "features_per_s_kernel": round(n / (k_ms * 1e-3), 0),
           "note": "latency bound: a feature is a dependent chain of ~6.0 k instructions in one lane, ~3.1 k with a 16-lane row "
                   "splitting the epipolar walk (rocprofv3 SQ_INSTS_VALU+SALU per wave, profiles/); images L2-resident, "
                   "80 B of HBM traffic per feature; the statistics counters are summed per wave into 64 spread slots (every "
                   "feature adding to the same words ran at 2.9 ns per feature: 24 us here, 171 us at 1080p)"}
    if not a.no_cpu_baseline:
        from oracle import stereo_capi as so

        frames = [dict(p, img_pad=so.make_frame(imgs[p["id"]], 5)[0]) for p in poses]
        newf = so.make_frame(imgs[12], 5)
        best = 1e9
        for _ in range(3):
            f = feats.copy().view(so.FEATURE_DTYPE)
            t0 = _t.perf_counter()
            so.update_feature_idepths(so.Params(), sc.K32, sc.Kinv32, w_, h_, 5, frames, newf, 11, f)
            best = min(best, _t.perf_counter() - t0)
        res["cpu_checker_us"] = round(best * 1e6, 1)
        res["cpu_checker_cores"] = 1
    return res


if __name__ == "__main__":
    main()
