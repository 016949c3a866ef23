"""Developer tool (GPU box, instrumented library): phase cycles of a short slab launch, one-workgroup kernel against the
two-role kernel with 1..3 row parts.  TLS_AMD_DEBUG=1 TLS_AMD_LIB=.../libtls_amd_clocks.so python tools/gpu_split_phases.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tls_amd import synthetic, _lib  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "tess_27d"
lo, hi = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (1000, 1307)
ctx = _lib.Context(0)
t, f, kw = synthetic.config(name)
inp = synthetic.search_inputs(t, f, **kw)
periods = inp["periods"][lo:hi]
for label, sw in (("one-wg", dict(split=0)), ("split", dict(split=1))):
    ctx.set_options(**sw)
    ctx.prepare(inp["t"], inp["y"], inp["dy"], periods, inp["table"], inp["params"])
    ctx.execute()
    ctx.synchronize()
    ms = ctx.execute_timed(3)
    ctx.execute(phase_clock=True)
    ph = ctx.phase_cycles()
    stats = {k: ph.pop(k) for k in list(ph) if k.startswith("stat_") or k.startswith("cumsum_")}
    tot = sum(ph.values())
    print("%-9s %.3f ms  cycles/period %.0f |" % (label, ms, tot / len(periods)),
          " ".join("%s=%.0f" % (k, v / len(periods)) for k, v in ph.items() if v >= 0.004 * tot), flush=True)
