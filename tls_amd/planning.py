"""What the batched search takes for one light curve under power(**kwargs): validated inputs, the period grid,
the template table and the scalar parameters -- the host-side planning power() itself does (main.py:51-123), as one
function for the callers that drive the C ABI directly (survey mode, the sharded search, benchmarks, tests)."""
import numpy


def search_inputs(t, flux, dy=None, **kwargs):
    """Everything the batched search takes for (t, flux[, dy]) under power(**kwargs):
    returns dict(t, y, dy, periods (ascending), table, params).  Host-side only."""
    from .api import transitleastsquares
    from .validate import validate_args

    model = transitleastsquares(t, flux, dy, verbose=False)
    kwargs = dict(kwargs)
    kwargs.setdefault("verbose", False)
    validate_args(model, kwargs)
    periods, durations, overview, rows, periods_sorted, table = model._build_grids()
    params = dict(transit_depth_min=model.transit_depth_min, R_star_min=model.R_star_min,
                  R_star_max=model.R_star_max, M_star_min=model.M_star_min,
                  M_star_max=model.M_star_max, T0_fit_margin=model.T0_fit_margin)
    return dict(t=model.t, y=model.y, dy=model.dy, periods=periods_sorted.copy(),
                table=table, params=params, overview=overview,
                rows=rows, durations=durations)
