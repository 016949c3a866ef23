"""Survey mode (BASELINE config 5): many light curves on the SAME time stamps, grids and
template searched back to back on one GPU by ONE C-ABI call (`tls_search_batch`).  The plan (period
list, duration windows, template rows, work queue order) is prepared once; per light curve only
the flux (and weights) are re-uploaded before the search kernel runs again.

Across GPUs the light curves are simply dealt out: one process per GPU (bench.py), or `devices=[0, 1, ...]` / `devices="auto"`
(every visible GPU) on the calls below -- contiguous slices of the batch to the contexts of a `tls_amd.search.DeviceGroup`, one host thread each, no
collective (every light curve's results come back over its own GPU's copy engine).
"""
import numpy

from . import search as _search
from .planning import search_inputs


def _batch_inputs(t, flux_batch, dy_batch, power_kwargs):
    """(inp, y_rows, dy_rows): the plan inputs of the first curve and every curve's (y, dy) exactly as
    validate.py would hand them to a single search (validate.py:18,39-40)."""
    flux_batch = numpy.asarray(flux_batch, dtype=numpy.float64)
    if flux_batch.ndim != 2 or flux_batch.shape[1] != len(t):
        raise ValueError("flux_batch must have shape [n_curves, len(t)]")
    first_dy = None if dy_batch is None else numpy.asarray(dy_batch[0], dtype=numpy.float64)
    inp = search_inputs(t, flux_batch[0], first_dy, **power_kwargs)
    if len(inp["t"]) != len(t):
        raise ValueError("light curves must be cleaned before a batched search")
    dy_rows = numpy.empty_like(flux_batch)
    dy_rows[0] = inp["dy"]
    for k in range(1, len(flux_batch)):
        if dy_batch is None:
            dy_rows[k] = numpy.std(flux_batch[k])
        else:
            dy = numpy.asarray(dy_batch[k], dtype=numpy.float64)
            dy_rows[k] = dy / numpy.mean(dy)
    y_rows = flux_batch.copy()
    y_rows[0] = inp["y"]
    return inp, y_rows, dy_rows


def _slices(n_curves, n_parts):
    """Contiguous, near-equal slices of the batch (whole launch groups of 32 where the batch allows it)."""
    groups = -(-n_curves // 32)
    bounds = [min(n_curves, 32 * ((groups * r) // n_parts)) for r in range(n_parts)] + [n_curves]
    if groups < n_parts:
        bounds = [(n_curves * r) // n_parts for r in range(n_parts)] + [n_curves]
    return bounds


def _resolve(devices, device, context, n_curves):
    """("group", DeviceGroup) or ("one", device id / None): search.resolve_devices, with "auto" = every visible GPU when
    the batch has at least two light curves for each."""
    kind, what = _search.resolve_devices(devices, device, context)
    if kind == "auto":
        from . import _lib
        n = _lib.device_count()
        kind, what = ("list", list(range(n))) if n > 1 and n_curves >= 2 * n else ("one", None)
    if kind == "list":
        kind, what = "group", _search.device_group(what)
    return kind, what


def _on_devices(group, call, n_curves):
    """call(context, lo, hi) for every device's slice of the batch, on the group's host threads; the slices' results."""
    with group._lock:   # (the group's contexts are not re-entrant: one batch or search at a time)
        bounds = _slices(n_curves, len(group.contexts))
        parts = [None] * len(group.contexts)

        def work(r):
            if bounds[r + 1] > bounds[r]:
                parts[r] = call(group.contexts[r], bounds[r], bounds[r + 1])

        group._threads(work)
    return [p for p in parts if p is not None]


def power_batch(t, flux_batch, dy_batch=None, context=None, device=None, with_arrays=False, devices=None, **power_kwargs):
    """Survey-mode power(): for every light curve of `flux_batch` what `transitleastsquares(t, flux).power(**kwargs)`
    reports as SDE, SDE_raw, chi2_min, period, T0, depth and duration (fractional, lc_cache_overview["duration"] of
    the template row at the chi^2 minimum, main.py:199-200) -- search, SDE spectra and final T0 fit all on the
    device (tls_power_batch), one record of 80 bytes back per light curve.

    Returns (summary, periods[, chi2, row, depth, power]): summary is a numpy structured array with the fields of
    tls_power_summary plus "duration".  The per-transit statistics of power() (SNR, odd/even, counts) are host work
    on a handful of candidates and are not part of the batch call."""
    inp, y_rows, dy_rows = _batch_inputs(t, flux_batch, dy_batch, power_kwargs)
    from . import constants as C
    osf = power_kwargs.get("oversampling_factor", C.OVERSAMPLING_FACTOR)
    kernel = osf * C.SDE_MEDIAN_KERNEL_SIZE
    if kernel != int(kernel):
        raise ValueError("oversampling_factor * %d must be an integer" % C.SDE_MEDIAN_KERNEL_SIZE)

    def call(ctx, lo, hi):
        return ctx.power_batch(inp["t"], y_rows[lo:hi], dy_rows[lo:hi], inp["periods"], inp["table"], inp["params"],
                               int(kernel), with_arrays=with_arrays, with_power=with_arrays)

    kind, what = _resolve(devices, device, context, len(y_rows))
    if kind == "group":
        parts = _on_devices(what, call, len(y_rows))
        raw = numpy.concatenate([p[0] for p in parts])
        chi2, row, depth, power = (numpy.concatenate([p[k] for p in parts]) if with_arrays else None for k in (1, 2, 3, 4))
    else:
        ctx = context if context is not None else _search.default_context(what)
        raw, chi2, row, depth, power = call(ctx, 0, len(y_rows))
    names = list(raw.dtype.names) + ["duration"]
    summary = numpy.zeros(len(raw), dtype=[(k, raw.dtype[k]) for k in raw.dtype.names] + [("duration", "f8")])
    for k in raw.dtype.names:
        summary[k] = raw[k]
    summary["duration"] = numpy.where(raw["no_fit"] != 0, numpy.nan, inp["table"].duration[raw["best_row"]])
    assert names == list(summary.dtype.names)
    if with_arrays:
        return summary, inp["periods"], chi2, row, depth, power
    return summary, inp["periods"]


def search_batch(t, flux_batch, dy_batch=None, context=None, device=None, devices=None, **power_kwargs):
    """Search every light curve of `flux_batch` (shape [n_curves, n_points]) on the grids that
    `transitleastsquares(t, flux).power(**power_kwargs)` would use.

    Returns (periods, chi2[n_curves, n_periods], row[...], depth[...]).  All light curves must
    share `t` and be free of invalid points (clean them first); a `dy_batch` must have the same
    weight structure for every curve (all uniform or all per-point).
    """
    inp, y_rows, dy_rows = _batch_inputs(t, flux_batch, dy_batch, power_kwargs)

    def call(ctx, lo, hi):
        return ctx.search_batch(inp["t"], y_rows[lo:hi], dy_rows[lo:hi], inp["periods"], inp["table"], inp["params"])

    kind, what = _resolve(devices, device, context, len(y_rows))
    if kind == "group":
        parts = _on_devices(what, call, len(y_rows))
        chi2, row, depth = (numpy.concatenate([p[k] for p in parts]) for k in range(3))
    else:
        ctx = context if context is not None else _search.default_context(what)
        chi2, row, depth = call(ctx, 0, len(y_rows))
    return inp["periods"], chi2, row, depth
