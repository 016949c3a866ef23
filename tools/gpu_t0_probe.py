"""Developer probe: wall time of tls_t0_fit against the number of trial epochs (512 workgroups: one sort each, then
rotations): the slope is what an epoch by rotation costs, the intercept the launch, the copies and the sort."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy
from tls_amd import synthetic, _lib
ctx = _lib.Context(0)
t, f, kw = synthetic.config("k2_90d")
period = 10.1245
sig = numpy.linspace(0.9999, 1.0, 70)
for n_ep in (1, 512, 1024, 2048, 4096, 8192, 16384):
    ep = numpy.linspace(t.min(), t.min() + period, n_ep)
    ctx.t0_fit_residuals(t, f, period, sig, ep, 36)
    best = 1e9
    for _ in range(20):
        t0 = time.perf_counter(); ctx.t0_fit_residuals(t, f, period, sig, ep, 36); best = min(best, time.perf_counter() - t0)
    print(n_ep, "epochs: %.1f us" % (best * 1e6), flush=True)
