"""Developer check on a GPU box: HIP search vs. the CPU oracle on synthetic configs."""
import sys
import time
import os

import numpy

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tls_amd import synthetic, _lib  # noqa: E402
import oracle  # noqa: E402


def run(name, t, flux, dy=None, oracle_stride=1, **kw):
    inp = synthetic.search_inputs(t, flux, dy, **kw)
    ctx = _lib.Context(0)
    p = inp["params"]
    t0 = time.time()
    chi2, row, depth, cnt = ctx.search(inp["t"], inp["y"], inp["dy"], inp["periods"], inp["table"], p, count_work=True)
    t1 = time.time()
    info = ctx.plan_info()
    ms = ctx.execute_timed(3)
    print(name, "N=%d periods=%d" % (len(inp["t"]), len(inp["periods"])), "first call %.3fs" % (t1 - t0),
          "kernel %.3f ms" % ms, "cells/s %.3e" % (cnt["grid_cells"] / (ms * 1e-3)), cnt, info, flush=True)
    sel = numpy.arange(0, len(inp["periods"]), oracle_stride)
    t2 = time.time()
    oc, orow, od, ocnt = oracle.search(inp["t"], inp["y"], inp["dy"], inp["periods"][sel], inp["table"],
                                       p["transit_depth_min"], p["R_star_min"], p["R_star_max"],
                                       p["M_star_min"], p["M_star_max"], p["T0_fit_margin"])
    t3 = time.time()
    rel = numpy.abs(chi2[sel] - oc) / oc
    print("   oracle %.2fs (%d periods, %d threads)" % (t3 - t2, len(sel), os.cpu_count()),
          "max rel dchi2 %.3e" % rel.max(), "rows equal", numpy.array_equal(row[sel], orow),
          "max ddepth %.3e" % numpy.abs(depth[sel] - od).max(),
          "argmin", int(numpy.argmin(chi2)), "chi2min %.10f" % chi2.min(), "oracle counters", ocnt, flush=True)
    bad = numpy.where(rel > 1e-9)[0]
    if len(bad):
        print("   worst:", [(int(sel[b]), chi2[sel[b]], oc[b]) for b in bad[:5]])
    ctx.close()


if __name__ == "__main__":
    print(_lib.Context(0).name)
    numpy.random.seed(0)
    t, f = synthetic.light_curve(30.0, 24, 2e-4, per=4.321, rp=0.05, a=12)
    run("small", t, f, period_min=3.5, period_max=5.5, oversampling_factor=2)
    t, f, kw = synthetic.config("k2_90d")
    run("k2_90d", t, f, **kw)
    dy = numpy.full(len(f), 5e-5); dy[::7] *= 2.0
    run("k2_90d+dy", t, f, dy, oracle_stride=10, **kw)
    t, f, kw = synthetic.config("tess_27d")
    run("tess_27d", t, f, oracle_stride=40, **kw)
