"""Developer tool (GPU box): the rotation path of the final T0 fit against the general kernel on random fits
(sizes, samplings, periods, template lengths, epochs anywhere, duplicate and nearly equal time stamps)."""
import os
import sys

import numpy

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tls_amd import _lib  # noqa: E402

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
rng = numpy.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
ctx = _lib.Context(0)
worst = 0.0
for case in range(n_cases):
    n = int(rng.choice([16, 17, 40, 100, 333, 1000, 2500, 6000]))
    kind = case % 5
    if kind == 0:
        t = numpy.linspace(2.0, 2.0 + float(rng.uniform(5, 60)), n)
    elif kind == 1:
        t = numpy.sort(rng.uniform(0, 40.0, n))
    elif kind == 2:   # nearly equal neighbours (gaps of a few ulps to 1e-12)
        t = numpy.sort(rng.uniform(0, 40.0, n))
        j = rng.randint(0, n - 1, size=max(1, n // 50))
        t[j + 1] = t[j] + rng.choice([0.0, 1e-15, 1e-13, 1e-11, 1e-9], size=len(j)) * rng.uniform(0.5, 2.0, len(j))
        t = numpy.sort(t)
    elif kind == 3:   # evenly sampled with a period commensurate (or nearly) with the cadence
        t = 1.0 + numpy.arange(n) / 48.0
    else:
        t = numpy.sort(numpy.concatenate([rng.uniform(0, 10, n // 2), rng.uniform(25, 30, n - n // 2)]))
    f = 1 + rng.normal(0, float(rng.choice([1e-5, 1e-3, 1e-2])), n)
    span = t.max() - t.min()
    if kind == 3:
        period = float(rng.choice([40 / 48.0, 41.5 / 48.0, 7.0, 77.3 / 48.0])) * float(rng.choice([1.0, 1.0 + 1e-9, 1.0 + 1e-6]))
    else:
        period = float(rng.uniform(0.05, 1.2) * span)
    dur = int(rng.randint(1, min(n, 300) + 1))
    signal = rng.uniform(0.98, 1.0, dur)
    lo = float(rng.choice([t.min(), t.min() - 3 * period, t.max(), t.min() + 1e3 * period]))
    epochs = numpy.sort(rng.uniform(lo, lo + period, int(rng.randint(1, 400))))
    roll = int(rng.choice([dur // 2 + 1, 0, 1, n - 1, n + 3]))
    ctx.set_options(t0_rot=None)
    got = ctx.t0_fit_residuals(t, f, period, signal, epochs, roll)
    ctx.set_options(t0_rot=0)
    want = ctx.t0_fit_residuals(t, f, period, signal, epochs, roll)
    rel = float(numpy.max(numpy.abs(got - want) / numpy.abs(want)))
    worst = max(worst, rel)
    if rel > 1e-12 or int(numpy.argmin(got)) != int(numpy.argmin(want)):
        print("MISMATCH case %d kind %d n %d period %r dur %d roll %d rel %.3g argmin %d %d" % (case, kind, n, period, dur, roll, rel, int(numpy.argmin(got)), int(numpy.argmin(want))), flush=True)
print("%d fits, worst relative difference %.3g" % (n_cases, worst))
