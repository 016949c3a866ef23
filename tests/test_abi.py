"""The C-ABI library: it loads without a GPU, exports every symbol the header
declares, its host-only planner matches the oracle's cell count, and it fails
loudly (no fallback) when no GPU is present."""
import os
import re

import numpy
import pytest

from tls_amd import _lib, synthetic
from conftest import REPO, oracle_search


def declared_functions():
    text = open(os.path.join(REPO, "include", "tls_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(tls_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    lib = _lib.load()
    names = declared_functions()
    assert len(names) >= 20
    for name in names:
        assert hasattr(lib, name), name
    assert sorted(_lib.SYMBOLS) == names
    assert lib.tls_version().startswith(b"tls_amd")


def test_no_gpu_means_error_not_fallback():
    lib = _lib.load()
    if lib.tls_device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(RuntimeError, match="GPU"):
        _lib.Context(0)


def test_host_planner_counts_cells_like_the_oracle(oracle_lib):
    t, f = synthetic.light_curve(20.0, 24, 3e-4, per=3.3, rp=0.05, a=10)
    for kw in ({}, {"T0_fit_margin": 0.1}, {"T0_fit_margin": 0}):
        inp = synthetic.search_inputs(t, f, period_min=1.0, period_max=6.0, **kw)
        sel = inp["periods"][::15]
        cells = _lib.grid_cells(inp["t"], sel, inp["table"], inp["params"])
        counters = oracle_search(oracle_lib, inp, periods=sel)[3]
        assert int(cells.sum()) == int(counters[0])
        assert numpy.all(cells > 0)
