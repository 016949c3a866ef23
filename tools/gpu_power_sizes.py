"""Developer tool (GPU box): wall time of the drop-in power() per configuration, and of its final T0 fit alone."""
import os
import sys
import time

import numpy

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tls_amd  # noqa: E402
from tls_amd import synthetic, _lib  # noqa: E402

names = sys.argv[1:] or ["k2_90d", "tess_27d", "kepler_4yr"]
for name in names:
    t, f, kw = synthetic.config(name)
    model = tls_amd.transitleastsquares(t, f, verbose=False)
    t0 = time.perf_counter()
    r = model.power(verbose=False, show_progress_bar=False, **kw)
    cold = time.perf_counter() - t0
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        r = model.power(verbose=False, show_progress_bar=False, **kw)
        best = min(best, time.perf_counter() - t0)
    print("%s: power() first %.3f s, then %.4f s; period %.5f T0 %.6f SDE %.3f, %d periods" % (name, cold, best, r.period, r.T0, r.SDE, len(r.periods)), flush=True)
    # the T0 fit alone, with the result's own parameters
    ctx = _lib.Context(0)
    dur = max(3, int(round(r.duration / numpy.median(numpy.diff(t)))))
    signal = numpy.ones(dur) - 1e-3
    n_epochs = min(len(t), int(len(t) / (0.01 * dur)))
    epochs = numpy.linspace(t.min(), t.min() + r.period, n_epochs)
    ctx.t0_fit_residuals(t, f, r.period, signal, epochs, dur // 2 + 1)
    t0 = time.perf_counter()
    ctx.t0_fit_residuals(t, f, r.period, signal, epochs, dur // 2 + 1)
    print("   T0 fit alone: dur %d, %d epochs: %.4f s" % (dur, n_epochs, time.perf_counter() - t0), flush=True)
