"""Developer tool: statistics of the fp32 screen (overlapping cells, cells valued in fp64) and one timed launch."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tls_amd import synthetic, _lib
name = sys.argv[1] if len(sys.argv) > 1 else "k2_90d"
t, f, kw = synthetic.config(name)
inp = synthetic.search_inputs(t, f, **kw)
ctx = _lib.Context(0)
ctx.prepare(inp["t"], inp["y"], inp["dy"], inp["periods"], inp["table"], inp["params"])
ctx.execute(); ctx.synchronize()
print("one launch: %.3f ms" % ctx.execute_timed(1), flush=True)
print("three launches: %.3f ms each" % ctx.execute_timed(3), flush=True)
ctx.execute(phase_clock=True)
ph = ctx.phase_cycles()
v = int(ph.get("stat_screen_parked", 0))
print("overlaps", v & 0xffffffff, "valued in batches", v >> 32, "valued at period end", ph.get("stat_screen_valued"), "periods", len(inp["periods"]))
