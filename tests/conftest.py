import json
import os
import sys

import numpy
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs an MI355X (run with `-m gpu` on the GPU box)")


class SimpleTable(object):
    """Template table as stored in the golden files (same fields as TemplateTable)."""

    def __init__(self, values, offset, length, width, overshoot):
        self.values, self.offset, self.length = values, offset, length
        self.width, self.overshoot = width, overshoot
        self.n_rows = len(width)


def load_search_golden(name):
    """Inputs/outputs of the reference's search_period (tools/gen_golden.py)."""
    g = numpy.load(os.path.join(GOLDEN, "search_%s.npz" % name))
    table = SimpleTable(g["tmpl_values"], g["tmpl_offset"], g["tmpl_length"], g["tmpl_width"],
                        g["tmpl_overshoot"])
    keys = ("transit_depth_min", "R_star_min", "R_star_max", "M_star_min", "M_star_max",
            "T0_fit_margin")
    params = dict(zip(keys, [float(v) for v in g["params"]]))
    return g, table, params


def load_power_golden(name):
    g = numpy.load(os.path.join(GOLDEN, "power_%s.npz" % name))
    kwargs = json.loads(str(g["kwargs_json"]))
    dy = g["in_dy"] if len(g["in_dy"]) else None
    return g, g["in_t"], g["in_y"], dy, kwargs


SEARCH_GOLDENS = ("small", "weights", "stride", "margin0", "nofit", "gap_ties")


@pytest.fixture(scope="session")
def oracle_lib():
    import oracle
    return oracle.OracleLibrary()


@pytest.fixture(scope="session")
def gpu():
    """A GPU context; fails (does not skip) when the HIP library or the GPU is missing."""
    from tls_amd import _lib
    ctx = _lib.Context(0)
    ctx.initial_options = ctx.get_options()   # (the process's TLS_* environment, read once: an A/B run of the suite keeps it)
    yield ctx
    # a checked build (make -C tls_amd/csrc debug, TLS_AMD_DEBUG=1 TLS_AMD_LIB=.../libtls_amd_debug.so) counts every violated
    # device-side bound: after the whole session none may have fired
    checked, counts = ctx.check_counts()
    ctx.close()
    if checked:
        print("\ndevice-side bound checks of the debug build:", counts)
        assert not any(counts.values()), counts


@pytest.fixture(autouse=True)
def _gpu_switches_back(request):
    """A test that changes the context's switches (gpu.set_options) leaves them as the session started with them."""
    yield
    if "gpu" in request.fixturenames:
        ctx = request.getfixturevalue("gpu")
        ctx.set_options(**ctx.initial_options)


def oracle_search(oracle_lib, inp, periods=None, n_threads=0):
    p = inp["params"]
    return oracle_lib.search(inp["t"], inp["y"], inp["dy"],
                             inp["periods"] if periods is None else periods, inp["table"],
                             p["transit_depth_min"], p["R_star_min"], p["R_star_max"],
                             p["M_star_min"], p["M_star_max"], p["T0_fit_margin"],
                             n_threads=n_threads)
