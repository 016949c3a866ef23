#!/usr/bin/env python
"""Benchmark of the TLS grid-search hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config k2_90d] [--mode survey|shard]

N > 1 is launched by `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N`
(one process per GPU; RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the environment).  Rank 0
prints ONE JSON line.

A "step" is one pass of the hot path over one batch of synthetic input that is already
resident in HBM: the full period x duration x T0 grid search of BASELINE.json's config 2 (90 d,
30-min cadence, default grids: 9679 periods, 8.77e8 trial cells).
  * survey mode (default; BASELINE config 5): every GPU searches one light curve per step
    (its own seed); the per-period (chi2, row, depth) triples of all K steps are exchanged with
    ONE RCCL all-gather at the end of the timed region, so that every rank holds the whole
    batch -> per-GPU work is fixed, "scaling": "weak".
  * shard mode (BASELINE config 4 layout): ONE light curve per step, its period grid block-
    partitioned over the GPUs by cumulative cell cost, one RCCL all-gather at the end
    -> "scaling": "strong".
value = trial cells of the whole job / wall time of the K timed steps (barrier + device sync on
both sides, max over ranks).  `roofline` prices the search kernel against HBM with the
algorithmic bytes of SURVEY.md 8(d) (24*N + 24 B per period) and its HIP-event duration, and
also reports the fp64 vector rate -- this path is compute/LDS bound, not HBM bound.
`cpu_baseline` is the C oracle (a port of core.py, OpenMP over periods) timed on this box's
host cores on a bounded sample of the same workload.
"""
import argparse
import json
import os
import sys
import time

import numpy

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from tls_amd import _lib, rendezvous, shard, synthetic  # noqa: E402

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
FP64_VECTOR_PEAK_TF = 78.6   # SURVEY.md section 7 (vector fp64, no MFMA on this path)


def cpu_baseline(inp, grid_cells, budget_s=20.0):
    """Oracle (port of the reference path) on the host cores, bounded sample.  SURVEY.md 8(d)
    also asks for a 1-core figure (the reference's FAQ quotes 4.3e6 cells/s/core) and a
    -ffast-math build (numba fastmath=True, core.py:28): both are reported beside `value`."""
    import oracle
    from oracle import oracle as oracle_build
    p = inp["params"]
    cores = oracle.usable_cores()   # affinity mask capped by the cgroup CPU quota
    periods = inp["periods"]

    def timed(lib, sel, n_threads):
        t0 = time.perf_counter()
        out = lib.search(inp["t"], inp["y"], inp["dy"], sel, inp["table"],
                         p["transit_depth_min"], p["R_star_min"], p["R_star_max"],
                         p["M_star_min"], p["M_star_max"], p["T0_fit_margin"], n_threads=n_threads)
        return time.perf_counter() - t0, int(out[3][0])

    def bounded(lib, n_threads, budget):
        width = cores if n_threads == 0 else n_threads
        probe = periods[:: max(1, len(periods) // (4 * width))]
        timed(lib, probe, n_threads)            # warms the OpenMP pool
        dt, cells = timed(lib, probe, n_threads)
        stride = max(1, int(numpy.ceil(dt * grid_cells / max(cells, 1) / budget)))
        sample = periods[::stride]
        dt, cells = timed(lib, sample, n_threads)
        if stride > 1 and dt < 0.4 * budget:    # the tiny probe overestimates (thread start-up)
            stride = max(1, int(numpy.ceil(dt * grid_cells / max(cells, 1) / budget)))
            sample = periods[::stride]
            dt, cells = timed(lib, sample, n_threads)
        reps = 1
        if stride == 1 and dt < 0.25 * budget:   # the whole grid is quick on this host: repeat it
            reps = max(1, min(50, int(0.5 * budget / dt)))
            t_all = 0.0
            for _ in range(reps):
                t_all += timed(lib, sample, n_threads)[0]
            dt = t_all / reps
        return cells / dt, "%d of %d periods (every %d%s) of the same light curve, %d x %.2f s" % (
            len(sample), len(periods), stride, "th" if stride > 1 else "st", reps, dt)

    strict = oracle.OracleLibrary()
    value, what = bounded(strict, 0, budget_s)
    out = {"value": value, "unit": "trial cells/s", "cores": cores, "kind": "port",
           "sample": what + ", oracle/tls_oracle.c -O2 OpenMP dynamic over periods; %d logical CPUs "
                     "visible, %d usable under the cgroup quota" % (os.cpu_count() or 1, cores)}
    one, what1 = bounded(strict, 1, 6.0)
    out["one_core"] = {"value": one, "sample": what1}
    try:  # -march=native: always compiled on the box that runs it
        oracle_build.build(fast=True, force=True)
        fast, whatf = bounded(oracle.OracleLibrary(fast=True), 0, 8.0)
        out["fastmath"] = {"value": fast, "cores": cores, "sample": whatf + ", -O3 -ffast-math -march=native"}
    except Exception as exc:  # the strict figure above stands on its own
        out["fastmath"] = {"value": None, "error": str(exc)[:200]}
    return out


RCCL_INIT_DEADLINE_S = 180.0


def _comm_init_with_deadline(ctx, world, rank, uid, deadline_s):
    """ncclCommInitRank through the C ABI on a helper thread (ctypes drops the GIL, every ABI call
    binds the context's device itself).  A rank that is still inside the call at the deadline
    reports failure, so that all ranks agree on the host-channel fallback instead of hanging the
    bench; the stuck thread is a daemon and main() leaves through os._exit in that case."""
    import threading
    outcome = {}

    def run():
        try:
            ctx.comm_init(world, rank, uid)
            outcome["ok"] = True
        except RuntimeError as exc:
            outcome["why"] = str(exc)

    th = threading.Thread(target=run, daemon=True)
    th.start()
    th.join(deadline_s)
    if th.is_alive():
        _STUCK.append(th)
        return False, "ncclCommInitRank did not return within %.0f s" % deadline_s
    return bool(outcome.get("ok")), outcome.get("why", "")


_STUCK = []


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default="k2_90d", choices=sorted(synthetic.CONFIGS))
    ap.add_argument("--sigma", type=float, default=None, help="noise override (e.g. 500e-6)")
    ap.add_argument("--mode", default="survey", choices=["survey", "shard"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the untimed extras (500 ppm variant, power() wall clock, counted pass) so "
                         "that a profiler sees only the timed workload's launches")
    ap.add_argument("--force-collective", action="store_true",
                    help="1-GPU runs: go through the RCCL code path with a one-rank communicator")
    args = ap.parse_args()

    rank, world, local_rank, addr, port = rendezvous.env_layout()
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus %d must be launched with torch.distributed.run "
                     "(--nproc-per-node %d)" % (args.gpus, args.gpus))
        args.gpus = world
    n_dev = _lib.device_count()
    ctx = _lib.Context(local_rank % max(n_dev, 1))
    channel = None
    collective = "none"
    if world > 1:
        # rank 0's RCCL unique id travels over the host channel; every rank then tries to join the
        # communicator.  If RCCL cannot be brought up on this box on ANY rank, all ranks fall back to
        # the host channel for the (tiny) result exchange and the output says so.
        channel = rendezvous.HostChannel(rank, world, addr, port)
        uid = channel.allgather_bytes(ctx.comm_unique_id() if rank == 0 else b"")[0]
        ok, why = _comm_init_with_deadline(ctx, world, rank, uid, RCCL_INIT_DEADLINE_S)
        if channel.all_true(ok):
            collective = "rccl"
        else:
            collective = "host-tcp fallback (RCCL init failed: %s)" % (why or "on another rank")
            if ok:
                ctx.comm_destroy()

    # ---- synthetic input, resident in HBM before the timed region --------------------
    seed = rank if args.mode == "survey" else 0
    t, flux, kw = synthetic.config(args.config, seed=seed, sigma=args.sigma)
    inp = synthetic.search_inputs(t, flux, **kw)
    periods = inp["periods"]
    if args.mode == "shard" and world > 1:
        job = shard.ShardedSearch(rank, world)
        lo, hi = job.plan(inp["t"], periods, inp["table"], inp["params"])
        my_periods = periods[lo:hi]
        count_per_rank = job.count_per_rank
        job_cells = int(numpy.sum(job.costs))
    else:
        my_periods = periods
        count_per_rank = len(periods)
        job_cells = None
    ctx.prepare(inp["t"], inp["y"], inp["dy"], my_periods, inp["table"], inp["params"])
    info = ctx.plan_info()
    if job_cells is None:
        job_cells = info["grid_cells"] * world  # survey: one full grid per GPU per step

    if world == 1 and args.force_collective:
        ctx.comm_init(1, 0, ctx.comm_unique_id())
        collective = "rccl"

    def barrier():
        if collective == "rccl":
            ctx.comm_barrier()
        elif channel is not None:
            channel.barrier()

    def reduce_max(v):
        if collective == "rccl":
            return ctx.comm_max(v)
        return channel.max(v) if channel is not None else v

    # Survey mode exchanges the results of all K steps with ONE all-gather at the end of the timed
    # region (north_star: "a single RCCL all-gather ... at the end"): per step the triples are only
    # parked in a slot of a device buffer, so no rank waits for another between two light curves.
    # Shard mode searches one light curve per step across all ranks, so its gather is per step.
    staged = collective == "rccl" and args.mode == "survey"
    n_slots = max(args.steps, 1)

    def step(i):
        ctx.execute()
        if staged:
            ctx.comm_stage_results(count_per_rank, i % n_slots, n_slots)
        elif collective == "rccl":
            # pack + ncclAllGather are enqueued behind the kernel: every rank holds the whole result
            # in HBM, and the next search starts without a host round trip
            ctx.comm_allgather_device(count_per_rank)
        elif channel is not None:
            c, r, d = ctx.fetch()
            channel.allgather_bytes(c.tobytes() + r.tobytes() + d.tobytes())

    def finish():
        if staged:
            ctx.comm_allgather_staged(count_per_rank, n_slots)

    if staged:   # every slot holds a result before the first gather ships the whole buffer
        for i in range(n_slots):
            step(i)
    for i in range(args.warmup):
        step(i)
    finish()
    ctx.synchronize()
    ctx.kernel_timing(reset=True)
    barrier()
    ctx.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i)
    finish()
    ctx.synchronize()
    barrier()
    ctx.synchronize()
    elapsed = time.perf_counter() - t0
    kernel_ms, launches = ctx.kernel_timing(reset=True)
    if staged:  # outside the timed region: one gathered slot on the host
        g_chi2, g_row, g_depth = ctx.comm_fetch_staged(count_per_rank, n_slots, n_slots - 1, world)
        assert len(g_chi2) == count_per_rank * world
    elif collective == "rccl":
        g_chi2, g_row, g_depth = ctx.comm_fetch_gathered(count_per_rank, world)
        assert len(g_chi2) == count_per_rank * world
    if world > 1:
        elapsed = reduce_max(elapsed)
        kernel_ms = reduce_max(kernel_ms)

    # one counted pass (outside the timed region) for the work statistics
    ctx.execute(count_work=True)
    chi2, row, depth, counters = ctx.fetch(with_counters=True)

    # secondary figure asked for by SURVEY.md 8(d): the same grid at 500 ppm noise, where 55 % of
    # the cells pass the depth predicate instead of 11 % (outside the timed region, rank 0 only)
    noisy = None
    if rank == 0 and args.sigma is None and args.config == "k2_90d" and not args.no_extras:
        t5, f5, kw5 = synthetic.config(args.config, seed=0, sigma=500e-6)
        i5 = synthetic.search_inputs(t5, f5, **kw5)
        ctx.prepare(i5["t"], i5["y"], i5["dy"], i5["periods"], i5["table"], i5["params"])
        ctx.execute(count_work=True)
        c5 = ctx.fetch(with_counters=True)[3]
        ms5 = ctx.execute_timed(5)
        noisy = {"sigma_ppm": 500.0, "kernel_ms": ms5, "trial_cells_per_s": c5["grid_cells"] / (ms5 * 1e-3),
                 "evaluated_fraction": c5["evaluated_cells"] / c5["grid_cells"],
                 "inner_steps": c5["inner_steps"]}

    # wall clock of the whole drop-in call for one light curve (host buffers in, results object
    # out: grids, template table, H2D, search, D2H, SDE spectra, device T0 fit, statistics)
    power_wall_ms = None
    if rank == 0 and not args.no_extras:
        import tls_amd
        model = tls_amd.transitleastsquares(t, flux, verbose=False)
        best = float("inf")
        for _ in range(4):
            t1 = time.perf_counter()
            model.power(verbose=False, show_progress_bar=False, context=ctx, **kw)
            best = min(best, time.perf_counter() - t1)
        power_wall_ms = 1e3 * best

    out = None
    if rank == 0:
        n = len(inp["t"])
        ms_per_step = 1e3 * elapsed / args.steps
        value = job_cells * args.steps / elapsed
        kernel_s = 1e-3 * kernel_ms / max(launches, 1)
        algo_bytes = len(my_periods) * (24 * n + 24)       # B_period per period
        achieved = algo_bytes / kernel_s / 1e9
        flops = 6.0 * counters["inner_steps"]               # reference flop count, core.py:68-69
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "hbm_traffic.json")
        if os.path.exists(tpath):
            rec = json.load(open(tpath))
            if rec.get("config") == args.config and rec.get("n_periods") == len(my_periods):
                traffic = rec.get("bytes_per_launch")
        out = {
            "metric": "trial cells/sec (period x duration x T0)",
            "value": value, "unit": "trial cells/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True,
            "scaling": "weak" if args.mode == "survey" else "strong",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "%s: %d points, %d periods x %d durations, %.3e trial cells per "
                                   "light curve; %s" % (
                                       args.config, n, len(periods), inp["table"].n_rows,
                                       info["grid_cells"] if args.mode == "survey" else job_cells,
                                       "survey mode, one light curve per GPU per step, one RCCL "
                                       "all-gather of all steps' results at the end" if args.mode == "survey" else
                                       "period grid sharded over the GPUs + RCCL all-gather"),
                       "mode": args.mode, "collective": collective, "sigma_ppm": 1e6 * (args.sigma or synthetic.CONFIGS[args.config][2]),
                       "light_curves_per_step": world if args.mode == "survey" else 1,
                       "search_ms_per_light_curve": ms_per_step / (world if args.mode == "survey" else 1),
                       "power_call_wall_ms_per_light_curve": power_wall_ms,
                       "evaluated_cells": counters["evaluated_cells"],
                       "inner_steps": counters["inner_steps"], "device": ctx.name,
                       "lds_bytes_per_workgroup": info["lds_bytes"], "workgroups": info["n_blocks"],
                       "lds_resident": info["resident"], "noisy_variant": noisy},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "kernel": "tls_search_kernel", "kernel_ms": 1e3 * kernel_s,
                         "algorithmic_bytes_per_launch": algo_bytes,
                         "note": "compute/LDS-bound path: HBM floor is 24*N+24 B per period; see fp64",
                         "fp64": {"achieved": flops / kernel_s / 1e12, "peak": FP64_VECTOR_PEAK_TF,
                                  "unit": "TFLOP/s", "frac": flops / kernel_s / 1e12 / FP64_VECTOR_PEAK_TF,
                                  "flops_per_launch": flops}},
            "argmin_period_index": int(numpy.argmin(chi2)), "chi2_min": float(numpy.min(chi2)),
        }
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(inp, info["grid_cells"])
    barrier()
    if collective == "rccl":
        ctx.comm_destroy()
    if not _STUCK:
        ctx.close()
    # RCCL writes a version banner through C stdio, which is flushed at exit when stdout is a
    # pipe.  Every rank flushes it now, then all ranks meet on the host channel, and only then
    # does rank 0 print: the JSON line is the LAST line of the job's combined output.
    try:
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except OSError:
        pass
    sys.stdout.flush()
    if channel is not None:
        channel.barrier()
    if rank == 0:
        print(json.dumps(out), flush=True)
    if channel is not None:
        channel.barrier()
        channel.close()
    if _STUCK:  # a helper thread is still inside RCCL: do not wait for it at interpreter exit
        sys.stderr.flush()
        os._exit(0)


if __name__ == "__main__":
    main()
