#!/usr/bin/env python
"""Benchmark of the TLS grid-search hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config k2_90d]

N > 1 runs one process per GPU: either under `python -m torch.distributed.run --nproc-per-node N
... bench.py --gpus N` (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the environment) or started
plainly as `python bench.py --gpus N`, in which case it spawns its N rank processes itself
(tls_amd/launch.py, the counterpart of the reference's Pool, main.py:140-163).  Rank 0 prints ONE
JSON line.

A "step" is one pass of the hot path over one batch of synthetic input that is already
resident in HBM: the full period x duration x T0 grid search of BASELINE.json's config 2 (90 d,
30-min cadence, default grids: 9679 periods, 8.77e8 trial cells).
  * `value` (survey layout, BASELINE config 5): every GPU searches one light curve per step (its
    own seed); the per-period (chi2, row, depth) triples of all K steps are exchanged with ONE
    RCCL all-gather at the end of the timed region, so that every rank holds the whole batch ->
    per-GPU work is fixed, "scaling": "weak".
  * N > 1 additionally times the PERIOD-SHARD layout north_star describes (BASELINE config 4): ONE
    light curve per step, its period grid dealt out over the GPUs (cyclic shares: rank r searches
    periods[r::N], tls_amd/shard.py), one RCCL all-gather per light curve -> strong scaling; reported
    under "shard" for config 2 and for the TESS 2-min configuration, with every rank's kernel time.
value = trial cells of the whole job / wall time of the K timed steps (barrier + device sync on
both sides, max over ranks), inputs resident in HBM when the timed region starts -- the bench contract
of this build ("if the boundary hands over host buffers, note the PCIe-inclusive rate ... it is never
`value`").  The search call FROM HOST BUFFERS (H2D + kernel + D2H through tls_search,
plan reused / planned from scratch) is printed beside it as `value_one_shot` / `value_one_shot_cold`.
`roofline` prices the dominant kernel of the headline configuration against the ceiling that binds it:
the LDS-resident kernel against the fp64 VECTOR rate (`bound: "fp64"`: one FMA per template tap of an
evaluated cell over its HIP-event duration; 78.6 TF peak) -- its light curve is L2-resident and the
algorithmic bytes of SURVEY.md 8(d) (24*N + 24 B per period) are nominal for it: they are reported
under `roofline.hbm` with the counter traffic --; the HBM-staged configurations (TESS 27 d, Kepler
4 yr: `tess_27d.roofline`, `kepler_4yr.roofline`) against HBM with those algorithmic bytes, the
counter traffic and the fp64 issue rate beside it.  At N = 1 the line also carries, measured outside
the timed region: those two configurations, the 1024-curve survey throughput including transfers (with
the slowest and the median group of 32), the noisier variants, the one-GPU projection of the 8-way
shard (`shard_balance`) and the wall clock of the whole power() call.
`cpu_baseline` is the C oracle (a port of core.py, OpenMP over periods) timed on this box's
host cores on a bounded sample of the same workload.
"""
import argparse
import json
import os
import subprocess
import sys
import time

import numpy

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from tls_amd import _lib, launch, rendezvous, shard, synthetic  # noqa: E402

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
           "sample": what + ", oracle/tls_oracle.c -O3 -ffp-contract=off OpenMP dynamic over periods; %d logical CPUs "
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


def kernel_sources_digest():
    """sha256 (16 hex digits) of the device + host sources of the library: what a traffic record was measured on."""
    import hashlib
    h = hashlib.sha256()
    base = os.path.join(ROOT, "tls_amd", "csrc")
    for name in sorted(os.listdir(base)):
        if name.endswith((".hip", ".h")):
            h.update(name.encode())
            h.update(open(os.path.join(base, name), "rb").read())
    return h.hexdigest()[:16]


_LIVE_TRAFFIC = {}   # config -> (bytes per launch, source) measured by this run (measure_traffic_live)


def measure_traffic_live(config, stride=1, timeout_s=120):
    """HBM bytes per launch of the search kernel measured IN THIS RUN: two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE: separate
    passes, kernel trace only -- MI355X_MICROARCH.md's HBM recipe) of a child process that prepares the configuration and
    launches its search a few times (tools/gpu_config_time.py); the read side doubled (gfx950 FETCH_SIZE counts 64 B per 128 B
    request), KiB -> bytes.  None when rocprofv3 is not there, the run is itself being profiled, or a pass fails or times
    out: recorded_traffic() then falls back to the committed record."""
    import csv
    import glob
    import shutil
    import tempfile
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe) or any(k.startswith(("ROCPROFILER_", "ROCPROF_", "ROCP_")) for k in os.environ) or \
            "rocprof" in os.environ.get("LD_PRELOAD", ""):
        return None   # (this run is itself under a profiler)
    child = [sys.executable, os.path.join(ROOT, "tools", "gpu_config_time.py"), config, str(stride), "6"]
    got = {}
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        out_dir = tempfile.mkdtemp(prefix="tls_pmc_", dir="/tmp")
        try:
            subprocess.run([exe, "--pmc", counter, "--output-format", "csv", "-d", out_dir, "-o", "k", "--"] + child, cwd="/tmp",
                           env=dict(os.environ, TMPDIR="/tmp"), timeout=timeout_s, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            values = []
            for path in glob.glob(os.path.join(out_dir, "**", "*counter_collection.csv"), recursive=True):
                for r in csv.DictReader(open(path)):
                    if r.get("Counter_Name") == counter and ("tls_search" in r.get("Kernel_Name", "") or "tls_slim" in r.get("Kernel_Name", "")
                                                              or "tls_fold_search" in r.get("Kernel_Name", "")):
                        values.append(float(r["Counter_Value"]))
            if not values:
                return None
            got[counter] = (sum(values) / len(values), len(values))
        except Exception:
            return None
        finally:
            shutil.rmtree(out_dir, ignore_errors=True)
    kib = 2.0 * got["FETCH_SIZE"][0] + got["WRITE_SIZE"][0]
    return kib * 1024.0, ("measured in this run: two rocprofv3 --pmc passes (FETCH_SIZE x2 + WRITE_SIZE, %d / %d launches) of "
                          "tools/gpu_config_time.py %s %d" % (got["FETCH_SIZE"][1], got["WRITE_SIZE"][1], config, stride))


def recorded_traffic(config, n_periods):
    """HBM bytes per launch: measured by this run where measure_traffic_live has been called for the configuration (N = 1,
    extras on), else from the committed rocprofv3 PMC passes (profiles/hbm_traffic.json: one record per configuration, each
    labelled with the commit and the kernel sources it was measured at)."""
    if config in _LIVE_TRAFFIC:
        b, source, measured_periods = _LIVE_TRAFFIC[config]
        scale = n_periods / float(measured_periods)
        return b * scale, source + ("" if scale == 1.0 else ", %d periods scaled to this launch's %d" % (measured_periods, n_periods))
    path = os.path.join(ROOT, "profiles", "hbm_traffic.json")
    if not os.path.exists(path):
        return None, "no profiles/hbm_traffic.json"
    rec = json.load(open(path))
    recs = rec.get("records", [rec] if "config" in rec else [])
    for r in recs:
        if r.get("config") == config and r.get("n_periods"):
            scale = n_periods / float(r["n_periods"])   # a period sample scales to the launch measured here
            # (the counter passes need rocprofv3 runs of their own: the record says which kernel sources it was taken on,
            # and the line says whether those are the sources this run was built from)
            current = r.get("kernel_sources") == kernel_sources_digest()
            return r.get("bytes_per_launch") * scale, "rocprofv3 PMC passes (%s) of commit %s, %d periods%s; kernel sources of the record %s" % (
                r.get("source", "FETCH_SIZE x2 + WRITE_SIZE"), r.get("commit", "unknown"), r["n_periods"],
                "" if scale == 1.0 else ", scaled by periods to this launch",
                "= this run's" if current else "DIFFER from this run's (%s vs %s): a stale figure" % (r.get("kernel_sources", "unrecorded"), kernel_sources_digest()))
    return None, "no record for %s" % config


class Harness(object):
    """One rank's view of the job: context, collective, barrier-bracketed timing."""

    def __init__(self, args):
        self.rank, self.world, local_rank, addr, port = rendezvous.env_layout()
        if self.world != args.gpus:   # a launcher's WORLD_SIZE wins over the flag
            args.gpus = self.world
        n_dev = _lib.device_count()
        self.ctx = _lib.Context(local_rank % max(n_dev, 1))
        self.channel = None
        self.devices = [local_rank % max(n_dev, 1)]
        self.collective = "none"
        if self.world > 1:
            # rank 0's RCCL unique id travels over the host channel; every rank then tries to join the
            # communicator.  If RCCL cannot be brought up on this box on ANY rank, all ranks fall back to
            # the host channel for the (tiny) result exchange and the output says so.
            self.channel = rendezvous.HostChannel(self.rank, self.world, addr, port)
            uid = self.channel.allgather_bytes(self.ctx.comm_unique_id() if self.rank == 0 else b"")[0]
            ok, why = _comm_init_with_deadline(self.ctx, self.world, self.rank, uid, RCCL_INIT_DEADLINE_S)
            import struct
            self.devices = [struct.unpack("<i", b)[0] for b in
                            self.channel.allgather_bytes(struct.pack("<i", local_rank % max(n_dev, 1)))]
            if self.channel.all_true(ok):
                self.collective = "rccl"
            elif len(set(self.devices)) == self.world and not args.allow_host_fallback:
                # every rank has a GPU of its own and RCCL still did not come up: that is a broken job, not a
                # configuration to time over TCP (the reference's pool silently running on one core would be a bug
                # there too, main.py:140-163).  All ranks leave with a non-zero status.
                sys.stderr.write("rank %d: RCCL communicator over %d distinct devices %s failed: %s\n"
                                 % (self.rank, self.world, self.devices, why or "on another rank"))
                sys.stderr.flush()
                self.channel.barrier()
                os._exit(3)
            else:
                # ranks share a device (a one-GPU box running the N-rank flow): RCCL refuses that by design
                self.collective = "host-tcp fallback (ranks share a device, RCCL init: %s)" % (why or "failed on another rank")
                if ok:
                    self.ctx.comm_destroy()
        elif args.force_collective:
            self.ctx.comm_init(1, 0, self.ctx.comm_unique_id())
            self.collective = "rccl"

    def barrier(self):
        if self.collective == "rccl":
            self.ctx.comm_barrier()
        elif self.channel is not None:
            self.channel.barrier()

    def reduce_max(self, v):
        if self.collective == "rccl":
            return self.ctx.comm_max(v)
        return self.channel.max(v) if self.channel is not None else v

    def timed(self, step, finish, steps, warmup, prefill=0, preroll=True):
        """W untimed + exactly K timed steps, barrier + device sync on both sides, max over ranks.
        Returns (seconds, mean search-kernel ms per launch from HIP events on the search stream)."""
        ctx = self.ctx
        for i in range(prefill):
            step(i)
        for i in range(warmup):
            step(i)
        finish()
        ctx.synchronize()
        # Untimed pre-roll: the shader clocks of a box that has just been handed over ramp for tens of launches (the
        # same kernel reads 2-3 % slower over the first ~25).  Plain searches, no collective (every rank on its own),
        # until two consecutive 10-launch means agree to 0.5 % -- so that K = 20 timed steps measure the steady state.
        self.preroll = {"batches": 0, "last_ms": None}
        if preroll:
            prev = None
            for b in range(40):
                ms = ctx.execute_timed(10)
                self.preroll = {"batches": b + 1, "last_ms": ms}
                if prev is not None and abs(ms - prev) <= 0.005 * prev:
                    break
                prev = ms
            ctx.synchronize()
        ctx.kernel_timing(reset=True)
        self.barrier()
        ctx.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            step(i)
        finish()
        ctx.synchronize()
        self.barrier()
        ctx.synchronize()
        elapsed = time.perf_counter() - t0
        kernel_ms, launches = ctx.kernel_timing(reset=True)
        kernel_ms = kernel_ms / max(launches, 1)
        self.own_kernel_ms = kernel_ms
        if self.world > 1:
            elapsed = self.reduce_max(elapsed)
            kernel_ms = self.reduce_max(kernel_ms)
        return elapsed, kernel_ms


def run_survey(h, args, inp):
    """Survey layout: one full-grid search of this rank's light curve per step."""
    ctx = h.ctx
    periods = inp["periods"]
    ctx.prepare(inp["t"], inp["y"], inp["dy"], periods, inp["table"], inp["params"])
    count_per_rank = len(periods)
    # the results of all K steps are exchanged with ONE all-gather at the end of the timed region
    # (north_star: "a single RCCL all-gather ... at the end"): per step the triples are only parked in
    # a slot of a device buffer, so no rank waits for another between two light curves
    staged = h.collective == "rccl"
    n_slots = max(args.steps, 1)

    def step(i):
        ctx.execute()
        if staged:
            ctx.comm_stage_results(count_per_rank, i % n_slots, n_slots)
        elif h.channel is not None:
            c, r, d = ctx.fetch()
            h.channel.allgather_bytes(c.tobytes() + r.tobytes() + d.tobytes())

    def finish():
        if staged:
            ctx.comm_allgather_staged(count_per_rank, n_slots)

    elapsed, kernel_ms = h.timed(step, finish, args.steps, args.warmup, prefill=n_slots if staged else 0)
    h.timed_kernel = ctx.last_kernel()   # which search kernel the timed steps launched (roofline.kernel)
    if staged:  # outside the timed region: one gathered slot on the host
        g_chi2, g_row, g_depth = ctx.comm_fetch_staged(count_per_rank, n_slots, n_slots - 1, h.world)
        assert len(g_chi2) == count_per_rank * h.world
    return elapsed, kernel_ms


def run_shard(h, args, config, steps=None):
    """Period-shard layout: ONE light curve per step, its period grid dealt out over the ranks (cyclic shares,
    tls_amd/shard.py: rank r searches periods[r::world]), one all-gather per light curve."""
    ctx = h.ctx
    t, flux, kw = synthetic.config(config, seed=0)
    inp = synthetic.search_inputs(t, flux, **kw)
    job = shard.ShardedSearch(h.rank, h.world)
    # (every rank derives the shares itself; plan() compares the digests of all ranks)
    mine = job.plan(inp["t"], inp["periods"], inp["table"], inp["params"], y=inp["y"], options=ctx.get_options(),
                    allgather_digests=h.channel.allgather_bytes if h.channel is not None else None)
    ctx.prepare(inp["t"], inp["y"], inp["dy"], numpy.ascontiguousarray(inp["periods"][mine]), inp["table"], inp["params"])
    c = job.count_per_rank

    def step(i):
        ctx.execute()
        if h.collective == "rccl":
            # pack + ncclAllGather are enqueued behind the kernel: every rank holds the whole result
            # in HBM, and the next search starts without a host round trip
            ctx.comm_allgather_device(c)
        elif h.channel is not None:
            cc, rr, dd = ctx.fetch()
            h.channel.allgather_bytes(cc.tobytes() + rr.tobytes() + dd.tobytes())

    n_steps = args.steps if steps is None else steps
    elapsed, kernel_ms = h.timed(step, lambda: None, n_steps, min(args.warmup, n_steps))
    if not getattr(h, 'timed_kernel', None):
        h.timed_kernel = ctx.last_kernel()
    own_kernel_ms = [h.own_kernel_ms]
    argmin = None
    if h.collective == "rccl":
        g = ctx.comm_fetch_gathered(c, h.world)
        chi2 = job.assemble(g[0])
        assert len(chi2) == len(inp["periods"])
        argmin = int(numpy.argmin(chi2))
    cells = numpy.array([numpy.sum(job.costs[job.indices(r)]) for r in range(h.world)], dtype=float)
    model = numpy.array([numpy.sum(job.times[job.indices(r)]) for r in range(h.world)], dtype=float)
    total = float(numpy.sum(job.costs))
    # measured balance: every rank's own kernel time per step (HIP events), gathered over the host channel
    kernel_ms_ranks = None
    if h.channel is not None:
        import struct
        parts = h.channel.allgather_bytes(struct.pack("<d", own_kernel_ms[0]))
        kernel_ms_ranks = [struct.unpack("<d", b)[0] for b in parts]
    return {"config": config, "points": len(inp["t"]), "periods": len(inp["periods"]), "trial_cells": total,
            "value": total * n_steps / elapsed, "unit": "trial cells/s", "scaling": "strong", "steps": n_steps,
            "ms_per_step": 1e3 * elapsed / n_steps, "kernel_ms_max_rank": kernel_ms,
            "cells_per_rank_max": float(cells.max()), "cells_per_rank_mean": float(cells.mean()),
            "cells_imbalance_max_over_mean": float(cells.max() / cells.mean()),
            "modelled_time_imbalance_max_over_mean": float(model.max() / model.mean()),
            "kernel_ms_per_rank": kernel_ms_ranks,
            "measured_time_imbalance_max_over_mean": (float(max(kernel_ms_ranks) / (sum(kernel_ms_ranks) / len(kernel_ms_ranks)))
                                                      if kernel_ms_ranks else None),
            "layout": job.layout, "periods_per_rank": [int(len(job.indices(r))) for r in range(h.world)],
            "argmin_period_index": argmin}


def noisy_variant(ctx, name, sigma, reps):
    """the same grid at a noise level where several times more cells pass the depth predicate."""
    t, flux, kw = synthetic.config(name, seed=0, sigma=sigma)
    inp = synthetic.search_inputs(t, flux, **kw)
    ctx.prepare(inp["t"], inp["y"], inp["dy"], inp["periods"], inp["table"], inp["params"])
    ctx.execute(count_work=True)
    c = ctx.fetch(with_counters=True)[3]
    ms = ctx.execute_timed(reps)
    return {"sigma_ppm": 1e6 * sigma, "kernel_ms": ms, "trial_cells_per_s": c["grid_cells"] / (ms * 1e-3),
            "evaluated_fraction": c["evaluated_cells"] / c["grid_cells"], "inner_steps": c["inner_steps"]}


def large_config(ctx, name, reps, warm=0):
    """Kernel time of one of the HBM-staged configurations (Kepler 4 yr, TESS 27 d) at its full
    grid, with its own HBM roofline: algorithmic bytes = periods x (24 N + 24) B."""
    t, flux, kw = synthetic.config(name, seed=0)
    inp = synthetic.search_inputs(t, flux, **kw)
    # first call of this size on the context: device and pinned buffers are (re)allocated (GBs of per-workgroup slabs
    # and lists); afterwards a new plan costs the host planning and one upload
    t0 = time.perf_counter()
    ctx.prepare(inp["t"], inp["y"], inp["dy"], inp["periods"][:-1], inp["table"], inp["params"])
    first_s = time.perf_counter() - t0
    ctx.synchronize()
    t0 = time.perf_counter()
    ctx.prepare(inp["t"], inp["y"], inp["dy"], inp["periods"], inp["table"], inp["params"])
    prep_s = time.perf_counter() - t0
    info = ctx.plan_info()
    ctx.execute()
    chi2 = ctx.fetch()[0]
    for _ in range(warm):   # (a short kernel right after host planning finds the clocks down: same ramp as the headline's)
        ctx.execute()
    ms = ctx.execute_timed(reps)
    n, n_per = len(inp["t"]), len(inp["periods"])
    algo = n_per * (24 * n + 24)
    achieved = algo / (ms * 1e-3) / 1e9
    traffic, source = recorded_traffic(name, n_per)
    # what the dot products of this launch come to against the fp64 vector rate (the slab variant's chi^2 phase is most of a
    # TESS-size period: the honest second ceiling beside HBM); one more launch, counting, outside the timed ones
    ctx.execute(count_work=True)
    c = ctx.fetch(with_counters=True)[3]
    fp64 = {"peak": FP64_VECTOR_PEAK_TF, "unit": "TFLOP/s", "useful": 2.0 * c["inner_steps"] / (ms * 1e-3) / 1e12,
            "issued": 2.0 * c["issued_fma"] / (ms * 1e-3) / 1e12,
            "useful_frac": 2.0 * c["inner_steps"] / (ms * 1e-3) / 1e12 / FP64_VECTOR_PEAK_TF,
            "issued_frac": 2.0 * c["issued_fma"] / (ms * 1e-3) / 1e12 / FP64_VECTOR_PEAK_TF,
            "lane_efficiency": c["inner_steps"] / max(c["issued_fma"], 1), "inner_steps": c["inner_steps"],
            "evaluated_fraction": c["evaluated_cells"] / max(c["grid_cells"], 1),
            "note": "one FMA per template tap of an evaluated cell (useful) / FMAs issued, over the kernel time of the plain launch"}
    # the whole drop-in call at this size (search + spectra + final T0 fit in one device submission, statistics on the host
    # with the pink noise on the device): the better of two calls after one that warms the caches
    power_ms = None
    try:
        import tls_amd
        model = tls_amd.transitleastsquares(t, flux, verbose=False)
        times = []
        for _ in range(3):
            t1 = time.perf_counter()
            model.power(verbose=False, show_progress_bar=False, context=ctx, **kw)
            times.append(time.perf_counter() - t1)
        power_ms = 1e3 * min(times[1:])
    except Exception:
        pass
    return {"points": n, "periods": n_per, "trial_cells": info["grid_cells"], "kernel_ms": ms, "fp64": fp64,
            "power_call_wall_ms": power_ms,
            "trial_cells_per_s": info["grid_cells"] / (ms * 1e-3), "host_prepare_ms": 1e3 * prep_s,
            "host_prepare_first_call_ms": 1e3 * first_s,
            "lds_resident": info["resident"], "argmin_period_index": int(numpy.argmin(chi2)),
            "best_period": float(inp["periods"][int(numpy.argmin(chi2))]),
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "algorithmic_bytes_per_launch": algo,
                         "traffic": traffic, "traffic_source": source,
                         # which ceiling is nearer: measured HBM traffic against 8 TB/s, or issued fp64 FMAs against 78.6 TF
                         "nearer_ceiling": ("fp64" if fp64["issued_frac"] > (traffic or algo) / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS else "hbm"),
                         "traffic_frac": (traffic or algo) / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "fp64_issued_frac": fp64["issued_frac"]}}


def shard_balance(ctx, name, n_blocks=8, reps=3):
    """The period-shard layout measured on ONE GPU: the grid cut into the `n_blocks` shares the ranks of an 8-GPU search
    would take (tls_amd/shard.py: cyclic, rank r searches periods[r::8]), every share searched alone -- what each rank
    would run, in the kernel a launch of that size takes -- and its kernel time taken with HIP events.  whole-grid time /
    slowest share = the speed-up 8 ranks would see (the all-gather of 24 B per period aside).  Beside it the layouts of
    earlier rounds: contiguous blocks placed by the fitted time model (rounds 3-5) and by trial cells alone (round 2)."""
    t, flux, kw = synthetic.config(name, seed=0)
    inp = synthetic.search_inputs(t, flux, **kw)
    out = {"config": name, "blocks": n_blocks, "periods": len(inp["periods"])}
    job = shard.ShardedSearch(0, n_blocks, layout="blocks")
    job.plan(inp["t"], inp["periods"], inp["table"], inp["params"], y=inp["y"], options=ctx.get_options())
    n_per = len(inp["periods"])
    layouts = (("cyclic", [shard.cyclic_indices(n_per, n_blocks, r) for r in range(n_blocks)]),
               ("time_model", [numpy.arange(job.bounds[r], job.bounds[r + 1]) for r in range(n_blocks)]),
               ("cells_only", None))
    for label, shares in layouts:
        if shares is None:
            b = shard.partition_by_cost(job.costs, n_blocks)
            shares = [numpy.arange(b[r], b[r + 1]) for r in range(n_blocks)]
        ms, kernels = [], set()
        for r in range(n_blocks):
            ctx.prepare(inp["t"], inp["y"], inp["dy"], numpy.ascontiguousarray(inp["periods"][shares[r]]), inp["table"], inp["params"])
            ctx.execute()
            ctx.synchronize()
            ms.append(ctx.execute_timed(reps))
            kernels.add(ctx.last_kernel())
        out[label] = {"periods_per_block": [int(len(ix)) for ix in shares], "kernel": sorted(kernels),
                      "kernel_ms_per_block": ms, "max_over_mean": max(ms) / (sum(ms) / len(ms)),
                      "slowest_block_ms": max(ms)}
    ctx.prepare(inp["t"], inp["y"], inp["dy"], inp["periods"], inp["table"], inp["params"])
    ctx.execute()
    ctx.synchronize()
    whole_ms = ctx.execute_timed(reps)
    out["whole_grid_kernel_ms"] = whole_ms
    out["layout"] = "cyclic"
    out["speedup_if_ranks_ran_the_blocks"] = whole_ms / out["cyclic"]["slowest_block_ms"]
    out["speedup_contiguous_time_model"] = whole_ms / out["time_model"]["slowest_block_ms"]
    return out


def survey_1024(ctx, n_curves):
    """BASELINE config 5 on one GPU: n_curves light curves (seeds 0..) of config 2 through ONE
    tls_search_batch call, host buffers in, host buffers out (transfers and host passes included)."""
    from tls_amd import survey
    t, f0, kw = synthetic.config("k2_90d", seed=0)
    fluxes = numpy.stack([synthetic.config("k2_90d", seed=s)[1] for s in range(n_curves)])
    survey.search_batch(t, fluxes[:64], context=ctx, **kw)   # warm: plan buffers, pinned staging
    t0 = time.perf_counter()
    periods, chi2, row, depth = survey.search_batch(t, fluxes, context=ctx, **kw)
    wall = time.perf_counter() - t0
    groups_search = ctx.batch_group_ms()
    best = numpy.argmin(chi2, axis=1)
    # survey-mode power(): search + SDE spectra + final T0 fit on the device, 80 bytes back per light curve
    survey.power_batch(t, fluxes[:64], context=ctx, **kw)
    t0 = time.perf_counter()
    summary, _ = survey.power_batch(t, fluxes, context=ctx, **kw)
    wall_power = time.perf_counter() - t0
    groups_power = ctx.batch_group_ms()

    def spread(g):   # (one stalled group of 32 -- r05's profile run held one call of 25.8 s among calls of 0.9 s -- shows as max >> median)
        return {"groups": int(len(g)), "median_ms": float(numpy.median(g)) if len(g) else None, "max_ms": float(numpy.max(g)) if len(g) else None,
                "argmax": int(numpy.argmax(g)) if len(g) else None}
    return {"curves": n_curves, "wall_s": wall, "curves_per_s": n_curves / wall,
            "ms_per_curve": 1e3 * wall / n_curves, "periods": len(periods),
            "argmin_seed0": int(best[0]), "note": "tls_search_batch, host buffers in and out",
            "curves_per_s_power": n_curves / wall_power, "power_wall_s": wall_power,
            "group_ms_search": spread(groups_search), "group_ms_power": spread(groups_power),
            "power_note": "tls_power_batch: per light curve SDE, SDE_raw, period, T0, depth, duration row, chi2_min "
                          "(search + spectra + final T0 fit on the device)",
            "seed0": {k: float(summary[0][k]) for k in ("SDE", "period", "T0", "depth", "duration")},
            "recovered_10_123d": int(numpy.sum(numpy.abs(summary["period"] - 10.123) < 0.05))}


def git_head():
    try:
        return subprocess.check_output(["git", "-C", ROOT, "rev-parse", "--short=12", "HEAD"],
                                       stderr=subprocess.DEVNULL).decode().strip()
    except Exception:
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # (defaults: a quarter of a second of steady state; with 3 + 20 steps the clocks are still ramping and the
    # same kernel reads 2.4 % slower)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--config", default="k2_90d", choices=sorted(synthetic.CONFIGS))
    ap.add_argument("--sigma", type=float, default=None, help="noise override (e.g. 500e-6)")
    ap.add_argument("--mode", default="both", choices=["both", "survey", "shard"],
                    help="N > 1: which layouts to time (value is always the survey layout when it runs)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the untimed extras (other configurations, 500 ppm variant, power() wall clock, "
                         "counted pass) so that a profiler sees only the timed workload's launches")
    ap.add_argument("--survey-curves", type=int, default=1024)
    ap.add_argument("--no-live-traffic", action="store_true",
                    help="do not run the rocprofv3 --pmc passes that measure roofline.traffic in this run (~10 s per configuration); "
                         "the committed record of profiles/hbm_traffic.json is reported instead")
    ap.add_argument("--allow-host-fallback", action="store_true",
                    help="N > 1 with one device per rank: time over the host channel when RCCL does not come up "
                         "(default: fail with a non-zero status)")
    ap.add_argument("--force-collective", action="store_true",
                    help="1-GPU runs: go through the RCCL code path with a one-rank communicator")
    args = ap.parse_args()

    # Started plainly (`python bench.py --gpus N`, no launcher): become the launcher.  N rank
    # processes of this same command line, one per GPU, with the environment contract of
    # torch.distributed.run (tls_amd/launch.py); rank 0's JSON line is this process's output.
    if args.gpus > 1 and not launch.launched_by_a_launcher():
        sys.exit(launch.spawn_ranks([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], args.gpus))

    h = Harness(args)
    ctx, rank, world = h.ctx, h.rank, h.world

    # ---- timed region: synthetic input resident in HBM before it starts ---------------
    t, flux, kw = synthetic.config(args.config, seed=rank, sigma=args.sigma)
    inp = synthetic.search_inputs(t, flux, **kw)
    periods = inp["periods"]
    elapsed = kernel_ms = None
    if args.mode in ("both", "survey") or world == 1:
        elapsed, kernel_ms = run_survey(h, args, inp)
    info = ctx.plan_info()
    shard_out = None
    if world > 1 and args.mode in ("both", "shard"):
        shard_out = [run_shard(h, args, args.config)]
        for other_cfg in ("tess_27d", "kepler_4yr"):   # BASELINE config 4, and config 3: the one grid with room to scale
            if args.config != other_cfg:
                shard_out.append(run_shard(h, args, other_cfg, steps=max(1, min(args.steps, 3)) if other_cfg == "kepler_4yr" else None))
        if elapsed is None:   # --mode shard: the first shard run is the headline
            elapsed = shard_out[0]["ms_per_step"] * 1e-3 * args.steps
            kernel_ms = shard_out[0]["kernel_ms_max_rank"]

    out = None
    if rank == 0:
        extras = world == 1 and not args.no_extras
        # one counted pass (outside the timed region) for the work statistics
        ctx.prepare(inp["t"], inp["y"], inp["dy"], periods, inp["table"], inp["params"])
        ctx.execute(count_work=True)
        chi2, row, depth, counters = ctx.fetch(with_counters=True)

        # the one-shot call a user of the C ABI sees -- host buffers in, host
        # buffers out: planning + H2D + kernel + D2H (tls_search), best of 5
        one_shot = None
        if extras:
            # cold: the context holds the plan of ANOTHER period grid, so tls_prepare plans from scratch (duration
            # windows, work order, template rows, one pinned upload).  warm: the same time stamps / grid / template
            # as the call before (a survey, repeated power() calls): the library recognises the plan and only the
            # flux travels.  Both: host buffers in, host buffers out.
            def one_call(per):
                t1 = time.perf_counter()
                ctx.prepare(inp["t"], inp["y"], inp["dy"], per, inp["table"], inp["params"])
                ctx.execute()
                ctx.fetch()
                return time.perf_counter() - t1
            cold = warm = float("inf")
            for _ in range(5):
                one_call(periods[:-1])          # evicts the plan
                cold = min(cold, one_call(periods))
                warm = min(warm, one_call(periods))
            one_shot = {"ms": 1e3 * warm, "trial_cells_per_s": info["grid_cells"] / warm,
                        "cold_ms": 1e3 * cold, "cold_trial_cells_per_s": info["grid_cells"] / cold,
                        "what": "tls_prepare + tls_execute + tls_fetch from host buffers (= tls_search): H2D, kernel, "
                                "D2H; `ms` with the plan of the previous call reused inside the library (same t, "
                                "periods, template: a survey), `cold_ms` with host planning from scratch"}

        # the same grid at 500 ppm noise, where 55 % of the cells pass the depth predicate instead of 11 %
        noisy = None
        if extras and args.sigma is None and args.config == "k2_90d":
            t5, f5, kw5 = synthetic.config(args.config, seed=0, sigma=500e-6)
            i5 = synthetic.search_inputs(t5, f5, **kw5)
            ctx.prepare(i5["t"], i5["y"], i5["dy"], i5["periods"], i5["table"], i5["params"])
            ctx.execute(count_work=True)
            c5 = ctx.fetch(with_counters=True)[3]
            ms5 = ctx.execute_timed(5)
            noisy = {"sigma_ppm": 500.0, "kernel_ms": ms5, "trial_cells_per_s": c5["grid_cells"] / (ms5 * 1e-3),
                     "evaluated_fraction": c5["evaluated_cells"] / c5["grid_cells"],
                     "inner_steps": c5["inner_steps"]}
            # ... and at 100 ppm, where the host takes the fp32-screen variant of the kernel (tls_amd.hip, screen_pays):
            # the three variants side by side (the context's switches, tls_options: prune / screen32)
            t1_, f1_, kw1_ = synthetic.config(args.config, seed=0, sigma=100e-6)
            i1_ = synthetic.search_inputs(t1_, f1_, **kw1_)
            variants = {}
            for label, sw in (("chosen_by_host", {"prune": None, "screen32": None}), ("plain", {"prune": 0, "screen32": 0}),
                              ("fp32_screen", {"prune": 0, "screen32": 1}), ("pruning", {"prune": 1, "screen32": 0})):
                ctx.set_options(**sw)
                ctx.prepare(i1_["t"], i1_["y"], i1_["dy"], i1_["periods"], i1_["table"], i1_["params"])
                ctx.execute(); ctx.synchronize()
                variants[label] = ctx.execute_timed(10)
            ctx.set_options(prune=None, screen32=None)
            noisy["at_100_ppm_kernel_ms"] = variants

        # wall clock of the whole drop-in call for one light curve (host buffers in, results object
        # out: grids, template table, H2D, search, D2H, SDE spectra, device T0 fit, statistics)
        power_wall_ms = None
        if extras:
            import tls_amd
            model = tls_amd.transitleastsquares(t, flux, verbose=False)
            best = float("inf")
            for _ in range(4):
                t1 = time.perf_counter()
                model.power(verbose=False, show_progress_bar=False, context=ctx, **kw)
                best = min(best, time.perf_counter() - t1)
            power_wall_ms = 1e3 * best

        # roofline.traffic measured in THIS run (N = 1): two rocprofv3 --pmc passes per configuration in a child process, while
        # this process idles (its context stays alive: the counters are per process)
        if extras and not args.no_live_traffic and args.sigma is None:
            for name, stride in ((args.config, 1),) + ((("tess_27d", 1), ("kepler_4yr", 64)) if args.config == "k2_90d" else ()):
                try:
                    live = measure_traffic_live(name, stride)
                    if live is not None:
                        t_l, f_l, kw_l = synthetic.config(name, seed=0)
                        n_per_l = len(inp["periods"]) if name == args.config else len(synthetic.search_inputs(t_l, f_l, **kw_l)["periods"])
                        _LIVE_TRAFFIC[name] = (live[0], live[1], len(range(0, n_per_l, stride)))
                except Exception:
                    pass

        other = {}
        if extras and args.config == "k2_90d":
            for name, reps, noisy_sigma in (("tess_27d", 30, 1000e-6), ("kepler_4yr", 2, 500e-6)):
                try:
                    other[name] = large_config(ctx, name, reps, warm=10 if name == "tess_27d" else 0)
                    other[name]["noisy_variant"] = noisy_variant(ctx, name, noisy_sigma, 1 if name == "kepler_4yr" else 3)
                except Exception as exc:
                    other.setdefault(name, {})["error"] = str(exc)[:300]
            try:
                other["survey_1024"] = survey_1024(ctx, args.survey_curves)
            except Exception as exc:
                other["survey_1024"] = {"error": str(exc)[:300]}
            other["shard_balance"] = []
            for name in ("k2_90d", "tess_27d", "kepler_4yr"):
                try:
                    other["shard_balance"].append(shard_balance(ctx, name, 8, 2 if name == "kepler_4yr" else 5))
                except Exception as exc:
                    other["shard_balance"].append({"config": name, "error": str(exc)[:300]})

        n = len(inp["t"])
        ms_per_step = 1e3 * elapsed / args.steps
        survey_ran = args.mode in ("both", "survey") or world == 1
        job_cells = info["grid_cells"] * world if survey_ran else shard_out[0]["trial_cells"]
        value = job_cells * args.steps / elapsed
        kernel_s = 1e-3 * kernel_ms
        algo_bytes = len(periods) * (24 * n + 24)          # B_period per period
        achieved = algo_bytes / kernel_s / 1e9
        ref_flops = 6.0 * counters["inner_steps"]           # reference flop count, core.py:68-69
        useful_flops = 2.0 * counters["inner_steps"]        # what the reformulated kernel needs: one FMA per tap
        issued_flops = 2.0 * counters["issued_fma"]         # what it issues (chunk padding, idle lanes, unroll slack)
        traffic, traffic_source = recorded_traffic(args.config, len(periods))
        out = {
            "metric": "trial cells/sec (period x duration x T0)",
            "value": value, "unit": "trial cells/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True,
            "scaling": "weak" if survey_ran else "strong",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            # the search call from host buffers (H2D + kernel + D2H); `_cold` with the host planning of a
            # first call, the other with the plan of the previous call reused inside the library (a survey)
            "value_one_shot": (one_shot or {}).get("trial_cells_per_s"),
            "value_one_shot_cold": (one_shot or {}).get("cold_trial_cells_per_s"),
            "config": {"workload": "%s: %d points, %d periods x %d durations, %.3e trial cells per "
                                   "light curve; %s" % (
                                       args.config, n, len(periods), inp["table"].n_rows, info["grid_cells"],
                                       "one light curve per GPU per step (inputs resident in HBM), one RCCL "
                                       "all-gather of all steps' results at the end" if survey_ran else
                                       "period grid sharded over the GPUs + RCCL all-gather"),
                       "mode": "survey" if survey_ran else "shard", "collective": h.collective,
                       "devices_per_rank": h.devices, "rccl_ranks": ctx.comm_ranks() if h.collective == "rccl" else 0,
                       "clock_preroll": getattr(h, "preroll", None),
                       "value_definition": "cells of the K timed steps / wall, inputs resident in HBM (tls_execute; the "
                                           "bench contract); value_one_shot is the search call from host buffers of "
                                           "SURVEY 8(d)(i) (tls_prepare + tls_execute + tls_fetch, plan reused), "
                                           "config.one_shot.cold_ms the same with host planning from scratch",
                       "sigma_ppm": 1e6 * (args.sigma or synthetic.CONFIGS[args.config][2]),
                       "light_curves_per_step": world if survey_ran else 1,
                       "search_ms_per_light_curve": ms_per_step / (world if survey_ran else 1),
                       "one_shot": one_shot,
                       "power_call_wall_ms_per_light_curve": power_wall_ms,
                       "evaluated_cells": counters["evaluated_cells"],
                       "inner_steps": counters["inner_steps"], "issued_fma": counters["issued_fma"],
                       "device": ctx.name, "commit": git_head(),
                       "lds_bytes_per_workgroup": info["lds_bytes"], "workgroups": info["n_blocks"],
                       "lds_resident": info["resident"], "noisy_variant": noisy},
            # What binds: the LDS-resident kernel (the light curve is 100 KB and L2-resident: measured HBM traffic is a
            # fraction of the algorithmic bytes) is priced against the fp64 VECTOR rate -- frac = useful FMAs (one per
            # template tap of an evaluated cell) over 78.6 TFLOP/s, half the guide's 157.3 TFLOP/s fp32 vector peak (fp64
            # issues at half rate: 4 cycles per wave64 instruction; no fp64 MFMA advantage on gfx950 and no GEMM here) --
            # with the HBM figure the contract asks for beside it.  The HBM-slab configurations below are priced against HBM.
            "roofline": ({"bound": "fp64", "achieved": useful_flops / kernel_s / 1e12, "peak": FP64_VECTOR_PEAK_TF, "unit": "TFLOP/s",
                          "frac": useful_flops / kernel_s / 1e12 / FP64_VECTOR_PEAK_TF} if info["resident"] else
                         {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS}),
            "argmin_period_index": int(numpy.argmin(chi2)), "chi2_min": float(numpy.min(chi2)),
        }
        out["roofline"].update({
            "traffic": traffic, "traffic_source": traffic_source,
            "kernel": "tls_slim_kernel" if getattr(h, "timed_kernel", "") == "slim" else "tls_search_kernel", "kernel_variant": getattr(h, "timed_kernel", ""),
            "kernel_ms": 1e3 * kernel_s,
            "algorithmic_bytes_per_launch": algo_bytes,
            "hbm": {"achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                    "note": "algorithmic bytes (24*N+24 B per period, SURVEY 8d) over the kernel time; nominal for the "
                            "LDS-resident kernel, whose measured HBM traffic (`traffic`) is a fraction of them"},
            "fp64": {"peak": FP64_VECTOR_PEAK_TF, "unit": "TFLOP/s",
                     "peak_source": "half of MI355X_MICROARCH.md's 157.3 TFLOP/s fp32 vector peak (256 CUs x 4 SIMD-32 x 2.4 GHz)",
                     "useful": useful_flops / kernel_s / 1e12,
                     "useful_frac": useful_flops / kernel_s / 1e12 / FP64_VECTOR_PEAK_TF,
                     "issued": issued_flops / kernel_s / 1e12,
                     "issued_frac": issued_flops / kernel_s / 1e12 / FP64_VECTOR_PEAK_TF,
                     "reference": ref_flops / kernel_s / 1e12,
                     "reference_frac": ref_flops / kernel_s / 1e12 / FP64_VECTOR_PEAK_TF,
                     "lane_efficiency": counters["inner_steps"] / max(counters["issued_fma"], 1),
                     "note": "useful: one FMA per template tap; issued: FMAs actually issued (chunk padding, idle lanes, "
                             "unroll slack); reference: the 6 flops per tap core.py:68-69 spends"}})
        out.update(other)
        if shard_out is not None:
            out["shard"] = shard_out
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(inp, info["grid_cells"])
    h.barrier()
    if h.collective == "rccl":
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
    if h.channel is not None:
        h.channel.barrier()
    if rank == 0:
        print(json.dumps(out), flush=True)
    if h.channel is not None:
        h.channel.barrier()
        h.channel.close()
    if _STUCK:  # a helper thread is still inside RCCL: do not wait for it at interpreter exit
        sys.stderr.flush()
        os._exit(0)


if __name__ == "__main__":
    main()
