"""Developer tool: self-time ranking of the host part of power() (cProfile, tottime)."""
import cProfile, os, pstats, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tls_amd
from tls_amd import synthetic, _lib
ctx = _lib.Context(0)
t, f, kw = synthetic.config("k2_90d")
m = tls_amd.transitleastsquares(t, f, verbose=False)
for _ in range(5):
    m.power(verbose=False, show_progress_bar=False, context=ctx, **kw)
pr = cProfile.Profile(); pr.enable()
for _ in range(50):
    m.power(verbose=False, show_progress_bar=False, context=ctx, **kw)
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(28)
st.sort_stats("cumulative").print_stats("tls_amd", 30)
