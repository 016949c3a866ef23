"""Developer probe: the LDS-resident kernel with other workgroup sizes and one or two workgroups per CU (tls_options
threads / blocks): kernel time and shader cycles per period (tls_debug_period_cycles: thread 0's clock around a period).
How a period's time depends on the waves that work on it and on the waves that share the CU."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy
from tls_amd import synthetic, _lib  # noqa: E402

ctx = _lib.Context(0)
t, f, kw = synthetic.config("k2_90d")
inp = synthetic.search_inputs(t, f, **kw)
for threads in (512, 448, 384, 320, 256, 192):
    for blocks in (512, 256):
        ctx.set_options(threads=threads, blocks=blocks)
        ctx.prepare(inp["t"], inp["y"], inp["dy"], inp["periods"], inp["table"], inp["params"])
        ctx.execute(); ctx.synchronize()
        ms = min(ctx.execute_timed(5) for _ in range(2))
        cyc = ctx.period_cycles().astype(float)
        chi2 = ctx.fetch()[0]
        print("threads %d (%d waves) x %d workgroups per CU: %.3f ms, %.0f cycles per period, chi2 min %.6f" % (
            threads, threads // 64, blocks // 256, ms, cyc.mean(), chi2.min()), flush=True)
