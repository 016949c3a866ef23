"""Developer probe: does a THIRD period slot per CU pay?  A 56-day light curve whose folded series fits the LDS three
times: two 512-thread workgroups per CU (what registers allow at 8 waves each) against three 320-thread ones (15 waves)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy
from tls_amd import synthetic, _lib
ctx = _lib.Context(0)
for span in (42.0,):
    t, f = synthetic.light_curve(span, 48, 50e-6)
    inp = synthetic.search_inputs(t, f)
    for threads, blocks in ((512, 512), (512, 1024), (256, 512), (256, 768), (256, 1024), (192, 1024), (192, 1280)):
        ctx.set_options(threads=threads, blocks=blocks)
        ctx.prepare(inp["t"], inp["y"], inp["dy"], inp["periods"], inp["table"], inp["params"])
        info = ctx.plan_info()
        ctx.execute(); ctx.synchronize()
        ms = min(ctx.execute_timed(10) for _ in range(3))
        print("span %.0f d, n %d, %d periods, lds %d B: threads %d, workgroups %d (asked %d): %.4f ms" % (
            span, len(inp["t"]), len(inp["periods"]), info["lds_bytes"], threads, info["n_blocks"], blocks, ms), flush=True)
