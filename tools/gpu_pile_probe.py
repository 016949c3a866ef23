"""Developer probe: phase clocks of the LDS-resident kernels on periods commensurate with the cadence (instrumented library)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy
from tls_amd import synthetic, _lib
ctx = _lib.Context(0)
n = 4320
t = 3.0 + numpy.arange(n) / 48.0
y = 1 + numpy.random.RandomState(5).normal(0, 5e-5, n)
inp = synthetic.search_inputs(t, y)
for P in (0.625, 2.5, 10.0, 7.123):
    periods = numpy.full(256, P)
    ctx.prepare(inp["t"], inp["y"], inp["dy"], periods, inp["table"], inp["params"])
    ctx.execute(); ctx.synchronize()
    ms = ctx.execute_timed(5)
    ctx.execute(phase_clock=True)
    ph = ctx.phase_cycles()
    tot = sum(v for k, v in ph.items() if not k.startswith("stat_") and not k.startswith("cumsum_"))
    print("P=%.3f %s %.4f ms cyc/period %.0f |" % (P, ctx.last_kernel(), ms, tot / 256.0),
          " ".join("%s=%.0f" % (k, v / 256.0) for k, v in ph.items() if v > 0.01 * tot or k == "stat_pruned_periods"), flush=True)
