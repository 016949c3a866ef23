"""Survey mode (BASELINE config 5): many light curves on the SAME time stamps, grids and
template searched back to back on one GPU by ONE C-ABI call (`tls_search_batch`).  The plan (period
list, duration windows, template rows, work queue order) is prepared once; per light curve only
the flux (and weights) are re-uploaded before the search kernel runs again.

Across GPUs the light curves are simply dealt to the ranks (one process per GPU); see bench.py.
"""
import numpy

from . import search as _search
from . import synthetic


def search_batch(t, flux_batch, dy_batch=None, context=None, device=None, **power_kwargs):
    """Search every light curve of `flux_batch` (shape [n_curves, n_points]) on the grids that
    `transitleastsquares(t, flux).power(**power_kwargs)` would use.

    Returns (periods, chi2[n_curves, n_periods], row[...], depth[...]).  All light curves must
    share `t` and be free of invalid points (clean them first); a `dy_batch` must have the same
    weight structure for every curve (all uniform or all per-point).
    """
    flux_batch = numpy.asarray(flux_batch, dtype=numpy.float64)
    if flux_batch.ndim != 2 or flux_batch.shape[1] != len(t):
        raise ValueError("flux_batch must have shape [n_curves, len(t)]")
    ctx = context if context is not None else _search.default_context(device)
    first_dy = None if dy_batch is None else numpy.asarray(dy_batch[0], dtype=numpy.float64)
    inp = synthetic.search_inputs(t, flux_batch[0], first_dy, **power_kwargs)
    if len(inp["t"]) != len(t):
        raise ValueError("light curves must be cleaned before a batched search")
    # dy exactly as validate.py would hand it to every single search (validate.py:18,39-40)
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
    chi2, row, depth = ctx.search_batch(inp["t"], y_rows, dy_rows, inp["periods"], inp["table"], inp["params"])
    return inp["periods"], chi2, row, depth
