"""Developer probe: the LDS-resident kernel with every CU holding ONE workgroup (blocks = CUs) against the default two --
per-phase shader clocks of both (instrumented library: make -C tls_amd/csrc clocks).  How much of a period's phases is
time waiting for the other workgroup's instructions, and how much is the period's own latency chain."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tls_amd import synthetic, _lib  # noqa: E402

ctx = _lib.Context(0)
t, f, kw = synthetic.config("k2_90d")
inp = synthetic.search_inputs(t, f, **kw)
for label, sw in (("2 WG/CU", {}), ("1 WG/CU", {"blocks": 256}), ("128 WGs", {"blocks": 128})):
    ctx.set_options(blocks=None)
    ctx.set_options(**sw)
    ctx.prepare(inp["t"], inp["y"], inp["dy"], inp["periods"], inp["table"], inp["params"])
    ctx.execute(); ctx.synchronize()
    ms = ctx.execute_timed(5)
    ctx.execute(phase_clock=True)
    ph = ctx.phase_cycles()
    ph = {k: v for k, v in ph.items() if not k.startswith("stat_") and not k.startswith("cumsum_")}
    tot = sum(ph.values())
    n = len(inp["periods"])
    print(label, "%.3f ms" % ms, "blocks", ctx.plan_info()["n_blocks"], "cyc/period %.0f |" % (tot / n),
          " ".join("%s=%.0f" % (k, v / n) for k, v in ph.items() if v >= 0.004 * tot), flush=True)
