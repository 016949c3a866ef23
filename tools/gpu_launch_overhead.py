"""Developer probe: kernel time against the number of periods of one launch (all of the same cost: the median period
repeated), LDS-resident configuration: the intercept is what a launch costs before and after its rounds of periods."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy
from tls_amd import synthetic, _lib
ctx = _lib.Context(0)
t, f, kw = synthetic.config(sys.argv[1] if len(sys.argv) > 1 else "k2_90d")
inp = synthetic.search_inputs(t, f, **kw)
for which, label in ((len(inp["periods"]) // 2, "median"), (len(inp["periods"]) - 1, "longest"), (0, "shortest")):
    P = inp["periods"][which]
    out = []
    for count in (1, 64, 256, 512, 513, 1024, 1536, 2048, 4096):
        periods = numpy.full(count, P)
        ctx.prepare(inp["t"], inp["y"], inp["dy"], periods, inp["table"], inp["params"])
        ctx.execute(); ctx.synchronize()
        ms = min(ctx.execute_timed(10) for _ in range(3))
        out.append("%d:%.4f" % (count, ms))
    print(label, "P=%.3f" % P, " ".join(out), flush=True)
