"""Developer tool (GPU box): which launches of the four-slot kernel stall?  Loops of survey.search_batch (batch launches alone),
survey.power_batch, and single searches; prints groups / calls beyond 10 x the median."""
import os
import sys
import time

import numpy

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tls_amd import synthetic, survey, _lib  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else "search_batch"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
sw = {}
for item in sys.argv[3:]:
    k, _, v = item.partition("=")
    sw[k] = float(v) if "." in v else int(v)
t, f0, kw = synthetic.config("k2_90d", seed=0)
fluxes = numpy.stack([synthetic.config("k2_90d", seed=s)[1] for s in range(1024)])
ctx = _lib.Context(0)
if sw:
    ctx.set_options(**sw)
inp = synthetic.search_inputs(t, f0, **kw)
bad_calls = 0
t_all = time.perf_counter()
if mode == "single":
    ctx.search(inp["t"], inp["y"], inp["dy"], inp["periods"], inp["table"], inp["params"])
    times = []
    for r in range(reps * 1024 // 8):
        ctx.update_flux(fluxes[r % 1024], numpy.full(len(t), numpy.std(fluxes[r % 1024])))
        t0 = time.perf_counter(); ctx.execute(); ctx.synchronize(); times.append(time.perf_counter() - t0)
    times = numpy.array(times)
    print("single searches: %d, median %.3f ms, max %.1f ms, beyond 10x: %d (%s)" % (len(times), 1e3 * numpy.median(times), 1e3 * times.max(),
          int(numpy.sum(times > 10 * numpy.median(times))), ctx.last_kernel()))
else:
    fn = survey.search_batch if mode == "search_batch" else survey.power_batch
    fn(t, fluxes[:64], context=ctx, **kw)
    for r in range(reps):
        t0 = time.perf_counter()
        fn(t, fluxes, context=ctx, **kw)
        wall = time.perf_counter() - t0
        g = ctx.batch_group_ms()
        if g.max() > 10 * numpy.median(g):
            bad_calls += 1
            print("rep %d wall %.2f s: group %d took %.1f s" % (r, wall, int(g.argmax()), g.max() / 1e3), flush=True)
    print("%s %s: %d of %d calls had a stalled group (%s), total %.0f s" % (mode, sw, bad_calls, reps, ctx.last_kernel(), time.perf_counter() - t_all))
