"""Developer A/B: slab fast mode with the dot products on X (default) against the re-staged samples (x_staged = 2),
per-phase clocks (instrumented library)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tls_amd import synthetic, _lib
ctx = _lib.Context(0)
for case in sys.argv[1:] or ["tess_27d", "kepler_4yr/64"]:
    name, _, stride = case.partition("/")
    t, f, kw = synthetic.config(name)
    inp = synthetic.search_inputs(t, f, **kw)
    periods = inp["periods"][::int(stride)] if stride else inp["periods"]
    for label, sw in (("x_dot", None), ("restage", 2)):
        ctx.set_options(x_staged=sw)
        ctx.prepare(inp["t"], inp["y"], inp["dy"], periods, inp["table"], inp["params"])
        ctx.execute(); ctx.synchronize()
        ms = min(ctx.execute_timed(5) for _ in range(3))
        ctx.execute(phase_clock=True)
        ph = ctx.phase_cycles()
        ph = {k: v for k, v in ph.items() if not k.startswith("cumsum_")}
        tot = sum(v for k, v in ph.items() if not k.startswith("stat_"))
        print(case, label, "%.3f ms" % ms, " ".join("%s=%d" % (k, v / len(periods)) for k, v in ph.items() if v >= 0.004 * tot or k.startswith("stat_")), flush=True)
