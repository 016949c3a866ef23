"""Developer tool: time a spread sample of the Kepler-size configuration (BASELINE config 3)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tls_amd import synthetic, _lib  # noqa: E402

ctx = _lib.Context(0)
t, f, kw = synthetic.config("kepler_4yr")
t0 = time.time()
inp = synthetic.search_inputs(t, f, **kw)
print("host setup %.2f s, %d periods, %d rows, W=%d" % (time.time() - t0, len(inp["periods"]),
      inp["table"].n_rows, inp["table"].width.max()), flush=True)
stride = int(sys.argv[1]) if len(sys.argv) > 1 else 64
sel = inp["periods"][::stride]
t0 = time.time()
ctx.prepare(inp["t"], inp["y"], inp["dy"], sel, inp["table"], inp["params"])
print("prepare %.2f s" % (time.time() - t0), ctx.plan_info(), flush=True)
ctx.execute(); ctx.synchronize()
ms = ctx.execute_timed(1)
info = ctx.plan_info()
print("kepler sample: %d periods, %.1f ms, cells %.3e -> %.3e cells/s; full grid estimate %.1f s"
      % (len(sel), ms, info["grid_cells"], info["grid_cells"] / ms * 1e3, ms * 1e-3 * stride), flush=True)
ctx.execute(phase_clock=True)
ph = ctx.phase_cycles()
tot = sum(v for k, v in ph.items() if not k.startswith("cumsum_"))
print({k: "%.1f%%" % (100.0 * v / tot) for k, v in ph.items() if not k.startswith("cumsum_")}, ph["cumsum_blocks"], ph["cumsum_fallbacks"])
