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
_groups_lock = threading.Lock()

# devices="auto" (the default of power(), validate.py:81's use_threads = cpu_count()): the grid goes over every visible
# GPU when the one-GPU search is modelled to take longer than sharding costs -- a launch and a share of one all-gather per
# extra rank (~50 us each, measured launch + small-message RCCL latency) -- and stays on one GPU otherwise
AUTO_OVERHEAD_S_PER_RANK = 50e-6
SHADER_CLOCK_HZ = 2.4e9   # tls_period_costs returns shader cycles of an MI355X


def resolve_devices(devices=None, device=None, context=None):
    """One place that reads the three ways a caller names GPUs (api.power, search_periods, survey.*_batch).

    Returns ("group", DeviceGroup) | ("list", [ids...]) (two or more) | ("one", device id or None) | ("auto", None).
    Raises ValueError for an empty list, non-integer ids, or devices= combined with device= / context=."""
    if devices is None:
        return "one", device
    if isinstance(devices, DeviceGroup):
        if device is not None or context is not None:
            raise ValueError("pass either devices= or device=/context=, not both")
        return "group", devices
    if isinstance(devices, str):
        if devices != "auto":
            raise ValueError('devices must be a list of GPU ids, a DeviceGroup or "auto"')
        if device is not None or context is not None:   # an explicit single device wins over the default "auto"
            return "one", device
        return "auto", None
    try:
        ids = [int(d) for d in devices]
        same = all(int(d) == d for d in devices)
    except (TypeError, ValueError):
        raise ValueError("devices must be a non-empty list of integer GPU ids")
    if not ids or not same or any(d < 0 for d in ids):
        raise ValueError("devices must be a non-empty list of non-negative integer GPU ids")
    if device is not None or context is not None:
        raise ValueError("pass either devices= or device=/context=, not both")
    if len(ids) == 1:
        return "one", ids[0]
    return "list", ids


def auto_devices(t, y, periods, table, params, n_visible=None, options=None):
    """devices="auto": the ids to shard over (a list of two or more) or None for one GPU.  The modelled one-GPU search
    time T1 (tls_period_costs: shader cycles per period, `slots` periods side by side) against the overhead of sharding:
    n GPUs are modelled to take T1 / n + AUTO_OVERHEAD_S_PER_RANK * (n - 1); the n with the smallest value is taken (every
    visible GPU once T1 is a few milliseconds; one GPU when the search is shorter than the overhead)."""
    from . import shard
    n_visible = _lib.device_count() if n_visible is None else int(n_visible)
    if n_visible <= 1 or len(periods) < 2 * n_visible:
        return None
    sigma = float(numpy.std(numpy.asarray(y, dtype=numpy.float64)))
    _, _, times, slots = _lib.period_costs(t, periods, table, params, sigma, with_slots=True, options=options)
    one_gpu_s = shard.block_makespan(times, slots) / SHADER_CLOCK_HZ
    modelled = [one_gpu_s / n + AUTO_OVERHEAD_S_PER_RANK * (n - 1) for n in range(1, n_visible + 1)]
    n_best = 1 + int(numpy.argmin(modelled))
    return list(range(n_best)) if n_best > 1 else None


def default_context(device=None):
    """Per-process context cache keyed by device id (one tls_ctx per GPU)."""
    dev = 0 if device is None else int(device)
    if dev not in _default_context:
        _default_context[dev] = _lib.Context(dev)
    return _default_context[dev]


class DeviceGroup(object):
    """The period grid of ONE search over several GPUs of this process: one tls_ctx and one host thread per listed
    device ("internally one host thread per GPU"), the periods dealt out cyclically (rank r takes
    periods[r::n], tls_amd.shard), and the per-period (chi2, row, depth) triples brought together by ONE collective: an RCCL
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
        self._lock = threading.RLock()   # one search (or batch) at a time: the contexts and the communicator are not re-entrant
        self.last_blocks = None          # periods per rank of the last search (cyclic shares: rank r took periods[r::n])
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
        try:
            self._threads(lambda r: self.contexts[r].comm_init(n, r, uid))   # ncclCommInitRank: every rank from its own thread
        except BaseException:
            self._drop()   # (a rank that never joined leaves the others' communicators half-built)
            raise
        self._comm_ready = True

    def _drop(self):
        """After a failure inside a collective the communicator is unusable: the group leaves the per-process cache and
        is closed (close() does both); the next call builds a fresh one."""
        try:
            self.close()
        except Exception:
            pass

    def search(self, t, y, dy, periods, table, params):
        with self._lock:
            return self._search(t, y, dy, periods, table, params)

    def _search(self, t, y, dy, periods, table, params):
        from . import shard
        if not self.contexts:
            raise RuntimeError("tls_amd: this DeviceGroup is closed")
        n = len(self.contexts)
        periods = numpy.ascontiguousarray(periods, dtype=numpy.float64)
        if n == 1:
            self.last_blocks, self.last_collective = numpy.asarray([0, len(periods)]), "none"
            return self.contexts[0].search(t, y, dy, periods, table, params)[:3]
        job = shard.ShardedSearch(0, n)
        job.plan(t, periods, table, params, with_costs=False)      # cyclic shares: rank r searches periods[r::n]
        count = job.count_per_rank
        shares = [job.indices(r) for r in range(n)]
        self.last_blocks = numpy.asarray([len(ix) for ix in shares])
        self._ensure_comm()
        parts = [None] * n

        # Two phases with a join between them: a rank that fails to prepare or execute (bad device, out of memory) must
        # not leave the others waiting in the all-gather for ever.  Phase 1 -- every rank searches its block and waits
        # for its own stream; only if ALL of them succeeded does phase 2 enter the collective.
        def search_block(r):
            ctx = self.contexts[r]
            ctx.prepare(t, y, dy, numpy.ascontiguousarray(periods[shares[r]]), table, params)
            ctx.execute()
            ctx.synchronize()

        self._threads(search_block)   # (raises before anybody has touched the communicator: it stays usable)

        def gather(r):
            ctx = self.contexts[r]
            # RCCL: every rank contributes its (zero-padded) block and receives all of them; otherwise its own block
            parts[r] = ctx.comm_allgather_results(count, n) if self.uses_rccl else ctx.fetch()

        try:
            self._threads(gather)
        except BaseException:
            if self.uses_rccl:
                self._drop()
            raise
        if self.uses_rccl:
            self.last_collective = "rccl_allgather"
            chi2, row, depth = parts[0]
            return job.assemble(chi2), job.assemble(row), job.assemble(depth)
        self.last_collective = "host_concatenate"
        out = []
        for k in range(3):
            full = numpy.empty(len(periods), dtype=parts[0][k].dtype)
            for r in range(n):
                full[shares[r]] = parts[r][k]
            out.append(full)
        return tuple(out)

    def close(self):
        with _groups_lock:   # (a closed group is never handed out again)
            for key, grp in list(_device_groups.items()):
                if grp is self:
                    del _device_groups[key]
        for ctx in self.contexts:
            if self._comm_ready:
                try:
                    ctx.comm_destroy()
                except Exception:
                    pass
            ctx.close()
        self.contexts = []
        self._comm_ready = False


def device_group(devices):
    """Per-process cache of DeviceGroups keyed by the device list (contexts and the RCCL communicator are kept)."""
    key = tuple(int(d) for d in devices)
    if not key:
        raise ValueError("devices must name at least one GPU")
    with _groups_lock:
        if key not in _device_groups:
            _device_groups[key] = DeviceGroup(key)
        return _device_groups[key]


def search_periods(t, y, dy, periods, table, transit_depth_min, R_star_min, R_star_max,
                   M_star_min, M_star_max, T0_fit_margin, context=None, device=None,
                   verbose=False, count_work=False, return_counters=False, devices=None, used=None):
    """chi2, row, depth for every trial period (same order as `periods`).

    table: tls_amd.template.TemplateTable.  Raises RuntimeError if the HIP
    library or a GPU is unavailable -- there is no CPU path.
    devices: a list of GPU ids shards the period grid over them (DeviceGroup); None / one id: one GPU; "auto": every
    visible GPU when the modelled one-GPU time exceeds the overhead of sharding (auto_devices).
    used: a dict that receives "context" (the context of the single device, or the group's first) and "devices".
    """
    params = dict(transit_depth_min=transit_depth_min, R_star_min=R_star_min,
                  R_star_max=R_star_max, M_star_min=M_star_min, M_star_max=M_star_max,
                  T0_fit_margin=T0_fit_margin)
    kind, what = resolve_devices(devices, device, context)
    if kind == "auto":
        ids = auto_devices(t, y, periods, table, params)
        kind, what = ("list", ids) if ids else ("one", None)
    if kind in ("group", "list"):
        group = what if kind == "group" else device_group(what)
        chi2, row, depth = group.search(t, y, dy, periods, table, params)
        if used is not None:
            used["context"], used["devices"] = group.contexts[0], list(group.devices)
        if verbose:
            print("GPU search on %d devices %s: periods per device %s, %s" % (len(group.devices), group.devices,
                                                                          list(group.last_blocks), group.last_collective))
        if return_counters:
            return chi2, row, depth, None
        return chi2, row, depth
    device = what
    ctx = context if context is not None else default_context(device)
    if used is not None:
        used["context"], used["devices"] = ctx, [getattr(ctx, "device", device)]
    chi2, row, depth, counters = ctx.search(t, y, dy, periods, table, params,
                                            count_work=count_work)
    if verbose:
        print("GPU search on " + ctx.name + ": " + str(counters["grid_cells"])
              + " trial cells")
    if return_counters:
        return chi2, row, depth, counters
    return chi2, row, depth


search_periods._tls_amd_product = True   # (api.power() takes the fused device chain only while this is the product's function)


def fused_power(t, y, dy, periods, table, params, oversampling_factor, context=None, device=None):
    """The device part of power() for ONE light curve on ONE GPU in a single submission (tls_power_batch with one light curve):
    search, SDE spectra (stats.py:105-132), the pick of main.py:198-212,269-272 and the final T0 fit (stats.py:135-204) -- trial
    epochs and scaled template formed on the device, first minimum taken on the device -- with ONE wait at the end instead of
    three (search fetch, spectra fetch, T0-fit fetch).  Returns (context, summary record, chi2, row, depth, SR, power_raw,
    power); every value equals what search_periods + spectra + final_T0_fit return for the same light curve."""
    from . import constants as C
    kernel = oversampling_factor * C.SDE_MEDIAN_KERNEL_SIZE
    if kernel != int(kernel):
        raise ValueError("oversampling_factor * %d must be an integer" % C.SDE_MEDIAN_KERNEL_SIZE)
    ctx = context if context is not None else default_context(device)
    y2 = numpy.ascontiguousarray(y, dtype=numpy.float64)[None, :]
    dy2 = numpy.ascontiguousarray(dy, dtype=numpy.float64)[None, :]
    summary, chi2, row, depth, power, SR, power_raw = ctx.power_batch(t, y2, dy2, periods, table, params, int(kernel),
                                                                      with_arrays=True, with_power=True, with_spectra=True)
    return ctx, summary[0], chi2[0], row[0], depth[0], SR[0], power_raw[0], power[0]


def t0_fit_residuals(t, y, period, signal, T0_array, roll, context=None, device=None):
    """Device evaluation of the final-T0-fit residuals (tls_t0_fit); same contract as
    tls_amd.stats.t0_fit_residuals_host."""
    ctx = context if context is not None else default_context(device)
    return ctx.t0_fit_residuals(t, y, period, signal, T0_array, roll)


# below this many window elements the numpy form of stats.pink_noise is quicker than a round trip to the device
PINK_NOISE_ON_DEVICE = 60000


def pink_noise(data, width, context=None, device=None):
    """stats.pink_noise (reference stats.py:72-77) on the device (tls_pink_noise): the same value bit for bit -- every
    window's two sums in numpy's pairwise association, the terms added one by one from the left.  Small inputs, and the
    inputs the reference itself fails on, stay with the numpy form."""
    from . import stats as _stats
    data = numpy.ascontiguousarray(data, dtype=float)
    width = int(width)
    n_windows = len(data) - width + 1
    if width < 1 or n_windows < 1 or n_windows * width < PINK_NOISE_ON_DEVICE or not numpy.all(numpy.isfinite(data)):
        return _stats.pink_noise(data, width)
    ctx = context if context is not None else default_context(device)
    return ctx.pink_noise(data, width)


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
