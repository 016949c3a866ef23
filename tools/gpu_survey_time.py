"""Developer tool: survey-mode throughput on one GPU (light curves per second)."""
import os
import sys
import time

import numpy

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tls_amd import synthetic, survey, _lib  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
t, f0, kw = synthetic.config("k2_90d", seed=0)
fluxes = numpy.stack([synthetic.config("k2_90d", seed=s)[1] for s in range(n)])
ctx = _lib.Context(0)
survey.search_batch(t, fluxes[:2], context=ctx, **kw)
t0 = time.perf_counter()
periods, chi2, row, depth = survey.search_batch(t, fluxes, context=ctx, **kw)
dt = time.perf_counter() - t0
print("%d light curves in %.3f s -> %.1f light curves/s, %.3e trial cells/s; 1024 would take %.2f s"
      % (n, dt, n / dt, n * ctx.plan_info()["grid_cells"] / dt, 1024 * dt / n))
