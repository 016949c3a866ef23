"""Batched period search on the GPU: the counterpart of the reference's
`search_period` mapped over the period grid (core.py:96-188, main.py:140-196).

`search_periods` is what `transitleastsquares.power()` calls.  It owns a default
per-process GPU context (created on first use) and accepts an explicit one.
"""
from . import _lib

_default_context = {}


def default_context(device=None):
    """Per-process context cache keyed by device id (one tls_ctx per GPU)."""
    dev = 0 if device is None else int(device)
    if dev not in _default_context:
        _default_context[dev] = _lib.Context(dev)
    return _default_context[dev]


def search_periods(t, y, dy, periods, table, transit_depth_min, R_star_min, R_star_max,
                   M_star_min, M_star_max, T0_fit_margin, context=None, device=None,
                   verbose=False, count_work=False, return_counters=False):
    """chi2, row, depth for every trial period (same order as `periods`).

    table: tls_amd.template.TemplateTable.  Raises RuntimeError if the HIP
    library or a GPU is unavailable -- there is no CPU path.
    """
    ctx = context if context is not None else default_context(device)
    params = dict(transit_depth_min=transit_depth_min, R_star_min=R_star_min,
                  R_star_max=R_star_max, M_star_min=M_star_min, M_star_max=M_star_max,
                  T0_fit_margin=T0_fit_margin)
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
