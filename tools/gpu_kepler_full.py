"""Developer tool: the full Kepler-size configuration (BASELINE config 3), search kernel and power()."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tls_amd  # noqa: E402
from tls_amd import synthetic, _lib  # noqa: E402

ctx = _lib.Context(0)
t, f, kw = synthetic.config("kepler_4yr")
inp = synthetic.search_inputs(t, f, **kw)
t0 = time.perf_counter()
chi2, row, depth, cnt = ctx.search(inp["t"], inp["y"], inp["dy"], inp["periods"], inp["table"], inp["params"], count_work=True)
dt = time.perf_counter() - t0
print("kepler full search: %d periods, N=%d: %.2f s wall (one-shot, counted), %.3e cells -> %.3e cells/s, evaluated %.3e, steps %.3e"
      % (len(chi2), len(inp["t"]), dt, cnt["grid_cells"], cnt["grid_cells"] / dt, cnt["evaluated_cells"], cnt["inner_steps"]), flush=True)
ms = ctx.execute_timed(1)
print("kernel only: %.1f ms -> %.3e cells/s; argmin period %.5f chi2_min %.4f" % (ms, cnt["grid_cells"] / ms * 1e3,
      inp["periods"][chi2.argmin()], chi2.min()), flush=True)
t0 = time.perf_counter()
r = tls_amd.transitleastsquares(t, f, verbose=False).power(verbose=False, show_progress_bar=False, context=ctx, **kw)
print("power() wall %.2f s: period %.5f SDE %.2f T0 %.5f depth %.6f" % (time.perf_counter() - t0, r.period, r.SDE, r.T0, r.depth), flush=True)
