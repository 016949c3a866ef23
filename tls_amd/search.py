"""Batched period search on the GPU: the counterpart of the reference's
`search_period` mapped over the period grid (core.py:96-188, main.py:140-196).

`search_periods` is what `transitleastsquares.power()` calls.  It owns a default
per-process GPU context (created on first use) and accepts an explicit one.
"""
import threading

import numpy

from . import _lib

_default_context = {}
_device_groups = {}


def default_context(device=None):
    """Per-process context cache keyed by device id (one tls_ctx per GPU)."""
    dev = 0 if device is None else int(device)
    if dev not in _default_context:
        _default_context[dev] = _lib.Context(dev)
    return _default_context[dev]


class DeviceGroup(object):
    """The period grid of ONE search over several GPUs of this process: one tls_ctx and one host thread per listed
    device ("internally one host thread per GPU"), contiguous period blocks placed by modelled time
    (tls_amd.shard), and the per-period (chi2, row, depth) triples brought together by ONE collective: an RCCL
    all-gather over the devices (tls_comm_*, ncclAllGather on every context's stream) when they are distinct GPUs,
    a device-to-host copy per block and a concatenation when the list names a GPU more than once (RCCL takes one
    rank per device).  The counterpart of the reference's Pool(processes=use_threads) over periods
    (main.py:140-163): a period's result does not depend on the partition (its prefix-sum mode is decided from the
    light curve and the period alone, tls_amd.hip enqueue), so the blocks return the bits of the one-device search.

    devices: device ids, e.g. [0, 1, 2, 3]; [0, 0] runs two contexts on GPU 0.  context_factory(device) -> an object
    with the methods of _lib.Context (tests pass a recording stand-in)."""

    def __init__(self, devices, context_factory=None):
        self.devices = [int(d) for d in devices]
        if not self.devices:
            raise ValueError("devices must name at least one GPU")
        factory = context_factory if context_factory is not None else _lib.Context
        self.contexts = [factory(d) for d in self.devices]
        self.distinct = len(set(self.devices)) == len(self.devices)
        self.uses_rccl = self.distinct and len(self.devices) > 1
        self._comm_ready = False
        self.last_blocks = None          # period-block boundaries of the last search
        self.last_collective = None      # "rccl_allgather" | "host_concatenate" | "none"

    def _threads(self, fn):
        """fn(rank) on one host thread per context; the first exception of any thread is raised here."""
        errors = []

        def run(r):
            try:
                fn(r)
            except BaseException as exc:   # (a failing rank must not leave the others waiting in a collective silently)
                errors.append((r, exc))

        pool = [threading.Thread(target=run, args=(r,), name="tls_amd-device-%d" % self.devices[r])
                for r in range(len(self.contexts))]
        for th in pool:
            th.start()
        for th in pool:
            th.join()
        if errors:
            r, exc = sorted(errors, key=lambda e: e[0])[0]
            raise RuntimeError("tls_amd: device %d (rank %d of %d) failed: %s" % (self.devices[r], r, len(self.contexts), exc)) from exc

    def _ensure_comm(self):
        if self._comm_ready or not self.uses_rccl:
            return
        uid = self.contexts[0].comm_unique_id()
        n = len(self.contexts)
        self._threads(lambda r: self.contexts[r].comm_init(n, r, uid))   # ncclCommInitRank: every rank from its own thread
        self._comm_ready = True

    def search(self, t, y, dy, periods, table, params):
        from . import shard
        n = len(self.contexts)
        periods = numpy.ascontiguousarray(periods, dtype=numpy.float64)
        if n == 1:
            self.last_blocks, self.last_collective = numpy.asarray([0, len(periods)]), "none"
            return self.contexts[0].search(t, y, dy, periods, table, params)[:3]
        job = shard.ShardedSearch(0, n)
        job.plan(t, periods, table, params, y=y, options=self.contexts[0].get_options())
        bounds, count = job.bounds, job.count_per_rank
        self.last_blocks = bounds
        self._ensure_comm()
        parts = [None] * n

        def work(r):
            ctx = self.contexts[r]
            lo, hi = int(bounds[r]), int(bounds[r + 1])
            ctx.prepare(t, y, dy, periods[lo:hi], table, params)
            ctx.execute()
            # RCCL: every rank contributes its (zero-padded) block and receives all of them; otherwise its own block
            parts[r] = ctx.comm_allgather_results(count, n) if self.uses_rccl else ctx.fetch()

        self._threads(work)
        if self.uses_rccl:
            self.last_collective = "rccl_allgather"
            chi2, row, depth = parts[0]
            return (shard.assemble(chi2, bounds, count), shard.assemble(row, bounds, count),
                    shard.assemble(depth, bounds, count))
        self.last_collective = "host_concatenate"
        return tuple(numpy.concatenate([parts[r][k] for r in range(n)]) for k in range(3))

    def close(self):
        for ctx in self.contexts:
            if self._comm_ready:
                try:
                    ctx.comm_destroy()
                except Exception:
                    pass
            ctx.close()
        self.contexts = []


def device_group(devices):
    """Per-process cache of DeviceGroups keyed by the device list (contexts and the RCCL communicator are kept)."""
    key = tuple(int(d) for d in devices)
    if key not in _device_groups:
        _device_groups[key] = DeviceGroup(key)
    return _device_groups[key]


def search_periods(t, y, dy, periods, table, transit_depth_min, R_star_min, R_star_max,
                   M_star_min, M_star_max, T0_fit_margin, context=None, device=None,
                   verbose=False, count_work=False, return_counters=False, devices=None):
    """chi2, row, depth for every trial period (same order as `periods`).

    table: tls_amd.template.TemplateTable.  Raises RuntimeError if the HIP
    library or a GPU is unavailable -- there is no CPU path.
    devices: a list of GPU ids shards the period grid over them (DeviceGroup); None / one id: one GPU.
    """
    params = dict(transit_depth_min=transit_depth_min, R_star_min=R_star_min,
                  R_star_max=R_star_max, M_star_min=M_star_min, M_star_max=M_star_max,
                  T0_fit_margin=T0_fit_margin)
    if devices is not None and (isinstance(devices, DeviceGroup) or len(devices) > 1):
        if context is not None:
            raise ValueError("pass either context= or devices=, not both")
        group = devices if isinstance(devices, DeviceGroup) else device_group(devices)
        chi2, row, depth = group.search(t, y, dy, periods, table, params)
        if verbose:
            print("GPU search on %d devices %s: period blocks %s, %s" % (len(group.devices), group.devices,
                                                                        list(numpy.diff(group.last_blocks)), group.last_collective))
        if return_counters:
            return chi2, row, depth, None
        return chi2, row, depth
    if devices is not None and len(devices) == 1 and device is None and context is None:
        device = list(devices)[0]
    ctx = context if context is not None else default_context(device)
    chi2, row, depth, counters = ctx.search(t, y, dy, periods, table, params,
                                            count_work=count_work)
    if verbose:
        print("GPU search on " + ctx.name + ": " + str(counters["grid_cells"])
              + " trial cells")
    if return_counters:
        return chi2, row, depth, counters
    return chi2, row, depth


def t0_fit_residuals(t, y, period, signal, T0_array, roll, context=None, device=None):
    """Device evaluation of the final-T0-fit residuals (tls_t0_fit); same contract as
    tls_amd.stats.t0_fit_residuals_host."""
    ctx = context if context is not None else default_context(device)
    return ctx.t0_fit_residuals(t, y, period, signal, T0_array, roll)


def spectra(chi2, oversampling_factor, context=None, device=None, resident=False):
    """SR, power_raw, power, SDE_raw, SDE from the chi^2 of every period (reference stats.py:105-132),
    evaluated on the device (tls_spectra: reductions + running-median detrend)."""
    # (search_periods is looked up at call time: a test that injects a CPU search also injects this function)
    from . import constants as C
    kernel = oversampling_factor * C.SDE_MEDIAN_KERNEL_SIZE
    if kernel != int(kernel):
        # (the reference indexes past the end of its window array for such a kernel, helpers.py:95-97)
        raise ValueError("oversampling_factor * %d must be an integer" % C.SDE_MEDIAN_KERNEL_SIZE)
    ctx = context if context is not None else default_context(device)
    # resident: read the chi2 the search left in HBM -- but only if `chi2` IS the array that search's fetch returned
    # and nothing has run on the context since (Context.holds); any other array (a patched search, an edited
    # copy, a second search in between) is uploaded
    return ctx.spectra(int(kernel), None if (resident and ctx.holds(chi2)) else chi2)
