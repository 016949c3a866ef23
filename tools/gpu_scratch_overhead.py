"""Developer probe: what a launch costs outside its periods -- kernel time (HIP events) of a ONE-period launch against the
shader cycles that period's workgroup counts for itself (tls_debug_period_cycles).  A kernel with scratch (spilled registers)
pays for it per launch."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy
from tls_amd import synthetic, _lib
ctx = _lib.Context(0)
for name in sys.argv[1:] or ["k2_90d", "tess_27d", "kepler_4yr"]:
    t, f, kw = synthetic.config(name)
    inp = synthetic.search_inputs(t, f, **kw)
    P = inp["periods"][len(inp["periods"]) // 2]
    for count in (1, 256):
        periods = numpy.full(count, P)
        ctx.prepare(inp["t"], inp["y"], inp["dy"], periods, inp["table"], inp["params"])
        ctx.execute(); ctx.synchronize()
        ms = min(ctx.execute_timed(10) for _ in range(3))
        cyc = numpy.median(ctx.period_cycles().astype(float))
        print(name, ctx.last_kernel(), "periods %d: kernel %.4f ms, a period's own cycles %.0f = %.4f ms at 2.4 GHz -> outside %.4f ms"
              % (count, ms, cyc, cyc / 2.4e6, ms - cyc / 2.4e6), flush=True)
