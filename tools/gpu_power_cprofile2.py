"""Developer tool (GPU box): cProfile of the drop-in power(), sorted by own time (200 calls)."""
import cProfile
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tls_amd  # noqa: E402
from tls_amd import synthetic  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "k2_90d"
t, f, kw = synthetic.config(name)
model = tls_amd.transitleastsquares(t, f, verbose=False)
for _ in range(5):
    model.power(verbose=False, show_progress_bar=False, **kw)
t0 = time.perf_counter()
for _ in range(100):
    model.power(verbose=False, show_progress_bar=False, **kw)
print("%s power() %.4f ms" % (name, 10 * (time.perf_counter() - t0)), flush=True)
pr = cProfile.Profile()
pr.enable()
for _ in range(200):
    model.power(verbose=False, show_progress_bar=False, **kw)
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(35)
