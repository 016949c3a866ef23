"""Developer tool: per-phase cycle breakdown of the search kernel on a GPU box."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tls_amd import synthetic, _lib  # noqa: E402

ctx = _lib.Context(0)
for name, sigma in (("k2_90d", None), ("k2_90d", 500e-6), ("tutorial01", None), ("tess_27d", None)):
    t, f, kw = synthetic.config(name, sigma=sigma)
    inp = synthetic.search_inputs(t, f, **kw)
    ctx.prepare(inp["t"], inp["y"], inp["dy"], inp["periods"], inp["table"], inp["params"])
    ctx.execute()
    ctx.synchronize()
    ms = ctx.execute_timed(5)
    ctx.execute(phase_clock=True)
    ph = ctx.phase_cycles()
    blocks, fails = ph.pop("cumsum_blocks"), ph.pop("cumsum_fallbacks")
    tot = sum(ph.values())
    info = ctx.plan_info()
    print(name, sigma, "%.3f ms" % ms, "cells/s %.3e" % (info["grid_cells"] / ms * 1e3),
          {k: "%.1f%%" % (100.0 * v / tot) for k, v in ph.items()},
          "cycles/period/wg %.0f" % (tot / len(inp["periods"])), "cumsum blocks %d fallbacks %d" % (blocks, fails), flush=True)
