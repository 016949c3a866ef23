"""Developer tool (GPU box): cProfile of the drop-in power() on one configuration."""
import cProfile
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tls_amd  # noqa: E402
from tls_amd import synthetic  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "tess_27d"
t, f, kw = synthetic.config(name)
model = tls_amd.transitleastsquares(t, f, verbose=False)
for _ in range(3):
    t0 = time.perf_counter()
    r = model.power(verbose=False, show_progress_bar=False, **kw)
    print("%s power() %.4f s" % (name, time.perf_counter() - t0), flush=True)
pr = cProfile.Profile()
pr.enable()
for _ in range(5):
    model.power(verbose=False, show_progress_bar=False, **kw)
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
