"""Developer tool (GPU box): series of 5-10 k points through the classic LDS-resident kernel (one 1024-thread workgroup per CU)
and the four-slot kernel's 512-thread shape (two workgroups per CU): kernel time, cells, rows, chi^2 distance."""
import os
import sys

import numpy

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tls_amd import synthetic, _lib  # noqa: E402

ctx = _lib.Context(0)
for span, cadence in ((110.0, 48), (150.0, 48), (180.0, 48), (54.0, 144)):
    t, f = synthetic.light_curve(span, cadence, 50e-6, per=10.123, rp=6371 / 696342, a=19)
    inp = synthetic.search_inputs(t, f)
    res = {}
    for label, sw in (("classic", dict(slim=0)), ("auto", dict(slim=None))):
        ctx.set_options(**sw)
        got = ctx.search(inp["t"], inp["y"], inp["dy"], inp["periods"], inp["table"], inp["params"], count_work=True)
        ctx.search(inp["t"], inp["y"], inp["dy"], inp["periods"], inp["table"], inp["params"])
        kern = ctx.last_kernel()
        for _ in range(30):
            ctx.execute()
        ctx.synchronize()
        ms = ctx.execute_timed(20)
        res[label] = (got, ms, kern)
    a, b = res["classic"][0], res["auto"][0]
    fin = numpy.isfinite(a[0])
    print("N=%d periods=%d  classic %.3f ms (%s)  auto %.3f ms (%s)  rows equal %s  cells %d/%d  max rel chi2 %.2e" % (
        len(t), len(inp["periods"]), res["classic"][1], res["classic"][2], res["auto"][1], res["auto"][2], numpy.array_equal(a[1], b[1]),
        a[3]["evaluated_cells"], b[3]["evaluated_cells"], numpy.max(numpy.abs(a[0][fin] - b[0][fin]) / numpy.abs(a[0][fin]))), flush=True)
