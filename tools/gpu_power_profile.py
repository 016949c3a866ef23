"""Developer tool: where the wall time of the drop-in power() call goes (cProfile on a GPU box)."""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tls_amd
from tls_amd import synthetic, _lib
ctx = _lib.Context(0)
t, f, kw = synthetic.config("k2_90d")
m = tls_amd.transitleastsquares(t, f, verbose=False)
for _ in range(3):
    m.power(verbose=False, show_progress_bar=False, context=ctx, **kw)
best = 1e9
for _ in range(10):
    t0 = time.perf_counter(); m.power(verbose=False, show_progress_bar=False, context=ctx, **kw); best = min(best, time.perf_counter() - t0)
print("power() best wall %.3f ms" % (best * 1e3))
pr = cProfile.Profile(); pr.enable()
for _ in range(20):
    m.power(verbose=False, show_progress_bar=False, context=ctx, **kw)
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(32)
pstats.Stats(pr).sort_stats("tottime").print_stats(28)
