"""Developer tool: wall-clock breakdown of the drop-in power() on a GPU box."""
import cProfile
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tls_amd  # noqa: E402
from tls_amd import synthetic  # noqa: E402

for name in ("k2_90d", "tutorial01"):
    t, f, kw = synthetic.config(name)
    model = tls_amd.transitleastsquares(t, f, verbose=False)
    model.power(verbose=False, show_progress_bar=False, **kw)  # warm-up (context, buffers)
    t0 = time.perf_counter()
    r = model.power(verbose=False, show_progress_bar=False, **kw)
    dt = time.perf_counter() - t0
    print(name, "power() wall %.3f s  period %.5f SDE %.3f" % (dt, r.period, r.SDE), flush=True)
    pr = cProfile.Profile()
    pr.enable()
    model.power(verbose=False, show_progress_bar=False, **kw)
    pr.disable()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(12)

# one-shot tls_search (host planning + H2D + kernel + D2H), the PCIe-inclusive rate of DESIGN.md
from tls_amd import _lib  # noqa: E402
ctx = _lib.Context(0)
t, f, kw = synthetic.config("k2_90d")
inp = synthetic.search_inputs(t, f, **kw)
ctx.search(inp["t"], inp["y"], inp["dy"], inp["periods"], inp["table"], inp["params"])
best = 1e9
for _ in range(5):
    t0 = time.perf_counter()
    ctx.search(inp["t"], inp["y"], inp["dy"], inp["periods"], inp["table"], inp["params"])
    best = min(best, time.perf_counter() - t0)
cells = ctx.plan_info()["grid_cells"]
print("one-shot tls_search k2_90d: %.3f ms -> %.3e cells/s (host buffers in and out)" % (1e3 * best, cells / best))
