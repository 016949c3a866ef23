"""Developer tool: survey-mode power() throughput on one GPU (tls_power_batch); run under rocprofv3 --kernel-trace --stats
to see which kernels the post-search part spends its time in."""
import os
import sys
import time

import numpy

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tls_amd import synthetic, survey, _lib  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
t, f0, kw = synthetic.config("k2_90d", seed=0)
fluxes = numpy.stack([synthetic.config("k2_90d", seed=s)[1] for s in range(n)])
ctx = _lib.Context(0)
survey.power_batch(t, fluxes[:32], context=ctx, **kw)
t0 = time.perf_counter()
summary, periods = survey.power_batch(t, fluxes, context=ctx, **kw)
dt = time.perf_counter() - t0
t0 = time.perf_counter()
survey.search_batch(t, fluxes, context=ctx, **kw)
ds = time.perf_counter() - t0
print("%d light curves: power_batch %.3f s (%.1f /s), search_batch %.3f s (%.1f /s)" % (n, dt, n / dt, ds, n / ds))
