"""Developer probe: phase clocks of single commensurate periods of a Kepler-size series (instrumented library)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy
from tls_amd import synthetic, _lib
n = 70128
t = 3.0 + numpy.arange(n) / 48.0
y = 1 + numpy.random.RandomState(5).normal(0, 5e-5, n)
inp = synthetic.search_inputs(t, y, period_min=0.5, period_max=400)
ctx = _lib.Context(0)
for P in (78 / 48.0, 66.5 / 48.0, 1.0, 131 / 48.0, 1.2345):
    periods = numpy.array([P])
    ctx.prepare(inp["t"], inp["y"], inp["dy"], periods, inp["table"], inp["params"])
    ctx.execute(); ctx.synchronize()
    ms = ctx.execute_timed(3)
    ctx.execute(phase_clock=True)
    ph = ctx.phase_cycles()
    ph = {k: v for k, v in ph.items() if v and not k.startswith("cumsum_")}
    print("P=%.4f %.3f ms" % (P, ms), " ".join("%s=%d" % kv for kv in ph.items()), flush=True)
