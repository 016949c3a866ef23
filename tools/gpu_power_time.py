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
