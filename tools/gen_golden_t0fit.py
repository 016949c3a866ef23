"""Generate tests/golden/t0fit_*.npz from the UNMODIFIED reference's final_T0_fit
(transitleastsquares/stats.py:135-204), build container only.

final_T0_fit returns only the best T0; the residual of every trial epoch lives in the local
variable `residuals_total` of its loop.  The reference is not edited: a sys.settrace hook reads
that local (and the trial grid `T0_array`, the rescaled `signal`) from the function's own frame
each time the loop reaches its comparison line (stats.py:199).  Everything stored is therefore an
input or a value computed by reference code.

Also stores spectra_*.npz: inputs/outputs of the reference's stats.spectra (stats.py:105-132).

Usage: python tools/gen_golden_t0fit.py
"""
import inspect
import os
import sys

import numpy

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, REPO)
import ref_shim  # noqa: E402

ref_shim.activate()
from transitleastsquares import stats as ref_stats  # noqa: E402
from transitleastsquares.transit import get_cache as ref_get_cache  # noqa: E402
from transitleastsquares import duration_grid as ref_duration_grid, period_grid as ref_period_grid  # noqa: E402
import transitleastsquares.tls_constants as C  # noqa: E402

from tls_amd import transit_model  # noqa: E402  (data generator only)

OUT = os.path.join(REPO, "tests", "golden")


def traced_final_T0_fit(**kwargs):
    """Run the reference function; return (T0, T0_array, residual per epoch, rescaled signal)."""
    fn = ref_stats.final_T0_fit
    src, first = inspect.getsourcelines(fn)
    cmp_line = first + [i for i, l in enumerate(src) if "if residuals_total < residuals_lowest" in l][0]
    rec = {"res": [], "T0_array": None, "signal": None}

    def local(frame, event, arg):
        if event == "line" and frame.f_lineno == cmp_line:
            rec["res"].append(float(frame.f_locals["residuals_total"]))
            rec["T0_array"] = frame.f_locals["T0_array"]
            rec["signal"] = frame.f_locals["signal"]
        return local

    def tracer(frame, event, arg):
        return local if frame.f_code is fn.__code__ else None

    sys.settrace(tracer)
    try:
        T0 = fn(**kwargs)
    finally:
        sys.settrace(None)
    return T0, numpy.array(rec["T0_array"], dtype=float), numpy.array(rec["res"]), numpy.array(rec["signal"], dtype=float)


def template_row(n, frac):
    """One row of the reference's own template table (transit.get_cache) of about frac*n samples."""
    periods = ref_period_grid(R_star=1, M_star=1, time_span=30.0, period_min=2.0, period_max=6.0, oversampling_factor=2)
    durations = ref_duration_grid(periods, shortest=1 / n, log_step=1.1)
    maxwidth = int(numpy.max(durations) * n)
    maxwidth += maxwidth % 2
    overview, rows = ref_get_cache(durations=durations, maxwidth_in_samples=maxwidth, per=C.DEFAULT_PERIOD,
                                   rp=C.DEFAULT_RP, a=C.DEFAULT_A, inc=C.DEFAULT_INC, ecc=C.DEFAULT_ECC,
                                   w=C.DEFAULT_W, u=C.DEFAULT_U, limb_dark=C.DEFAULT_LIMB_DARK, verbose=False)
    k = int(numpy.argmin(numpy.abs(overview["width_in_samples"] - frac * n)))
    return numpy.asarray(rows[k], dtype=float)


def case(name, t, y, dy, period, depth, margin, frac):
    signal = template_row(len(t), frac)
    T0, T0_array, res, scaled = traced_final_T0_fit(signal=signal, depth=depth, t=t, y=y, dy=dy, period=period,
                                                    T0_fit_margin=margin, show_progress_bar=False, verbose=False)
    assert len(res) == len(T0_array)
    numpy.savez_compressed(os.path.join(OUT, "t0fit_%s.npz" % name), t=t, y=y, dy=dy, period=period, depth=depth,
                           T0_fit_margin=margin, signal=signal, T0=T0, T0_array=T0_array, residuals=res,
                           scaled_signal=scaled)
    print("t0fit_%s: N=%d dur=%d epochs=%d T0=%.10f res_min=%.10f" % (name, len(t), len(signal), len(res), T0, res.min()))


def main():
    numpy.random.seed(0)
    n = 720
    t = numpy.linspace(3.14, 33.14, n)
    y = transit_model.light_curve(t, 4.14, 4.321, 0.05, 12, 90, 0, 90, [0.4, 0.4], "quadratic") + numpy.random.normal(0, 2e-4, n)
    dy = numpy.full(n, numpy.std(y))
    case("small", t, y, dy, 4.3234, 0.9975, 0.01, 0.02)
    case("margin0", t, y, dy, 4.3234, 0.9975, 0.0, 0.03)
    case("coarse", t, y, dy, 2.5, 0.999, 0.1, 0.05)
    # unsorted time stamps with exact ties, per-point dy (the reference overwrites dy: no influence)
    p = numpy.random.permutation(n)
    t2, y2 = t[p].copy(), y[p].copy()
    t2[10:14] = t2[10]
    t2[500] = t2[20]
    case("ties", t2, y2, numpy.random.uniform(1e-4, 4e-4, n), 4.3234, 0.998, 0.01, 0.02)

    # spectra (stats.py:105-132): a chi2 array with a dip, long enough for the median detrend
    for name, n_p, osf in (("detrended", 400, 3), ("short", 150, 3), ("os5", 700, 5)):
        numpy.random.seed(7)
        chi2 = 1000 + numpy.cumsum(numpy.random.normal(0, 0.3, n_p)) + numpy.random.normal(0, 1.0, n_p)
        chi2[n_p // 3] -= 40
        chi2[n_p // 3 + 1] -= 25
        SR, power_raw, power, SDE_raw, SDE = ref_stats.spectra(chi2, osf)
        numpy.savez_compressed(os.path.join(OUT, "spectra_%s.npz" % name), chi2=chi2, oversampling_factor=osf, SR=SR,
                               power_raw=power_raw, power=power, SDE_raw=SDE_raw, SDE=SDE)
        print("spectra_%s: n=%d SDE_raw=%.8f SDE=%.8f" % (name, n_p, SDE_raw, SDE))


if __name__ == "__main__":
    main()
