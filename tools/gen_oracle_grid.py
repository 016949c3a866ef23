#!/usr/bin/env python
"""Whole-grid output of the CPU ORACLE (oracle/tls_oracle.c) for a benchmark configuration,
stored as a fixture so that the GPU box need not spend ~2 core-hours on it at test time.

    python tools/gen_oracle_grid.py kepler_4yr [n_threads]

Writes tests/golden/oracle_<config>_grid.npz: chi2 (f8), row (i2), depth (f8) for EVERY period of
the default grid, plus the oracle's counters.  These are ORACLE outputs, not reference outputs --
the oracle itself is pinned against the unmodified reference by tests/test_oracle_golden.py
(tools/gen_golden.py); this fixture only saves run time.  tests/test_gpu_parity.py re-checks a
>= 1000-period sample of it against the live oracle before trusting it.
"""
import os
import sys
import time

import numpy

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import oracle  # noqa: E402
from tls_amd import synthetic  # noqa: E402


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "kepler_4yr"
    n_threads = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    t, f, kw = synthetic.config(name)
    inp = synthetic.search_inputs(t, f, **kw)
    p = inp["params"]
    lib = oracle.OracleLibrary()
    periods = inp["periods"]
    chi2 = numpy.empty(len(periods))
    row = numpy.empty(len(periods), dtype=numpy.int64)
    depth = numpy.empty(len(periods))
    counters = numpy.zeros(3, dtype=numpy.int64)
    t0 = time.time()
    block = 8192
    for lo in range(0, len(periods), block):
        hi = min(lo + block, len(periods))
        c, r, d, cnt = lib.search(inp["t"], inp["y"], inp["dy"], periods[lo:hi], inp["table"],
                                  p["transit_depth_min"], p["R_star_min"], p["R_star_max"],
                                  p["M_star_min"], p["M_star_max"], p["T0_fit_margin"], n_threads=n_threads)
        chi2[lo:hi], row[lo:hi], depth[lo:hi] = c, r, d
        counters += cnt
        print("%d / %d periods, %.0f s" % (hi, len(periods), time.time() - t0), flush=True)
    out = os.path.join(ROOT, "tests", "golden", "oracle_%s_grid.npz" % name)
    assert row.max() < 32768
    numpy.savez_compressed(out, chi2=chi2, row=row.astype(numpy.int16), depth=depth, counters=counters,
                           n_points=len(inp["t"]), periods_first_last=periods[[0, -1]])
    print("wrote", out, os.path.getsize(out), "bytes; argmin", int(numpy.argmin(chi2)), chi2.min())


if __name__ == "__main__":
    main()
