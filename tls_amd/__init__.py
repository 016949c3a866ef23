"""tls_amd -- Transit Least Squares with the grid search on AMD MI355X (HIP).

Drop-in for the public surface of `transitleastsquares` (reference __init__.py:13-18):
    from tls_amd import transitleastsquares, period_grid, duration_grid, ...
`catalog_info` (network catalogue queries) is out of scope and not provided.
"""
from .api import transitleastsquares  # noqa: F401
from .helpers import cleaned_array, resample, transit_mask, fold  # noqa: F401
from .grid import duration_grid, period_grid  # noqa: F401
from .stats import FAP  # noqa: F401
from .results import transitleastsquaresresults  # noqa: F401
from .constants import VERSION as __version__  # noqa: F401
