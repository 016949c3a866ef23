"""Developer tool (GPU box): hunt the rare multi-second stall of one group of tls_power_batch (round 5: one call of 25.8 s;
round 6: caught by the guard test, group 15 of 32 took 25.75 s).  Repeats the 1024-curve call and prints every group beyond
10 x the median with the part of its time spent waiting for the device."""
import os
import sys
import time

import numpy

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tls_amd import synthetic, survey, _lib  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
with_oracle = len(sys.argv) > 2 and sys.argv[2] == "oracle"
t, f0, kw = synthetic.config("k2_90d", seed=0)
fluxes = numpy.stack([synthetic.config("k2_90d", seed=s)[1] for s in range(1024)])
ctx = _lib.Context(0)
survey.power_batch(t, fluxes[:64], context=ctx, **kw)
if with_oracle:   # (the test session runs the OpenMP oracle in the same process before this call)
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import oracle
    inp = synthetic.search_inputs(t, f0, **kw)
    p = inp["params"]
    oracle.search(inp["t"], inp["y"], inp["dy"], inp["periods"][::4], inp["table"], p["transit_depth_min"], p["R_star_min"], p["R_star_max"],
                  p["M_star_min"], p["M_star_max"], p["T0_fit_margin"])
worst = 0.0
for r in range(reps):
    t0 = time.perf_counter()
    survey.power_batch(t, fluxes, context=ctx, **kw)
    wall = time.perf_counter() - t0
    g, w = ctx.batch_group_ms(with_wait=True)
    med = numpy.median(g)
    bad = numpy.nonzero(g > 10 * med)[0]
    worst = max(worst, g.max())
    if len(bad) or r % 10 == 0:
        print("rep %d wall %.3f s median group %.1f ms max %.1f ms" % (r, wall, med, g.max()),
              " ".join("group %d: %.1f ms (wait %.1f)" % (i, g[i], w[i]) for i in bad), flush=True)
print("worst group over %d calls: %.1f ms" % (reps, worst))
