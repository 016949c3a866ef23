"""Post-search statistics on the host: the final T0 fit (its loop on the device), transit
times, per-transit depth/SNR/count statistics, FAP lookup.

Host numpy.  Behaviour follows the reference's stats.py:10-469 including the
quirks listed in SURVEY.md Appendix C (15: the T0 fit weights by 1/flux^2).
The T0 fit -- the second-hottest loop of a search (stats.py:178-201) -- is
evaluated here in batches of trial T0s with one stable argsort per batch.
"""
import os

import numpy

from . import constants as C
from .helpers import fold, running_median, transit_mask

_FAP_CACHE = []


def _fap_table():
    """(FAP[k], SDE threshold[k]) rebuilt from the stored SDE thresholds; the FAP
    column is analytic (tools/gen_fap_table.py)."""
    if not _FAP_CACHE:
        milli = numpy.load(os.path.join(C.DATA_DIR, "fap_sde_milli.npy"))
        thr = numpy.append(milli.astype(numpy.float64) / 1000.0, numpy.inf)
        k = numpy.arange(len(thr))
        fap = numpy.round(numpy.maximum(len(thr) - 1 - k, 1) / 12495.0, 9)
        fap[0] = numpy.nan
        _FAP_CACHE.append((fap, thr))
    return _FAP_CACHE[0]


def FAP(SDE):
    """False-alarm probability of a detection with the given SDE (white-noise
    simulations of the reference authors; reference stats.py:10-18)."""
    fap, thr = _fap_table()
    return fap[numpy.argmax(thr > SDE)]


def rp_rs_from_depth(depth, law, params):
    """Planet-to-star radius ratio from the maximum transit depth for a given
    limb-darkening law (Heller 2019, arXiv:1901.01730; reference stats.py:21-69)."""
    if len(params) == 1:
        params = float(params[0])
    if not isinstance(params, (float, int)) and not all(
            isinstance(x, (float, int)) for x in params):
        raise ValueError("All limb-darkening parameters must be numbers")
    laws = "linear, quadratic, squareroot, logarithmic, nonlinear"
    if law not in laws:
        raise ValueError("Please provide a supported limb-darkening law:", laws)
    if law == "linear" and not isinstance(params, float):
        raise ValueError("Please provide exactly one parameter")
    if law in "quadratic, logarithmic, squareroot" and len(params) != 2:
        raise ValueError("Please provide exactly two limb-darkening parameters")
    if law == "nonlinear" and len(params) != 4:
        raise ValueError("Please provide exactly four limb-darkening parameters")

    if law == "linear":
        factor = 1 - params / 3
    elif law == "quadratic":
        factor = 1 - params[0] / 3 - params[1] / 6
    elif law == "squareroot":
        factor = 1 - params[0] / 3 - params[1] / 5
    elif law == "logarithmic":
        factor = 1 + 2 * params[1] / 9 - params[0] / 3
    else:
        factor = 1 - params[0] / 5 - params[1] / 3 - 3 * params[2] / 7 - params[3] / 2
    return (depth * factor) ** (1 / 2)


def pink_noise(data, width):
    """Mean over all windows of `width` points of std(window)/sqrt(width)
    (reference stats.py:72-77).  Window statistics in one vectorised call; the
    running total is accumulated left to right like the reference's loop."""
    data = numpy.asarray(data, dtype=float)
    n_windows = len(data) - width + 1
    windows = numpy.lib.stride_tricks.sliding_window_view(data, width)
    terms = numpy.std(windows, axis=1) / width ** 0.5
    total = 0
    for v in terms.tolist():
        total += v
    return total / n_windows


def period_uncertainty(periods, power):
    """Half of the full width at half maximum of the highest power peak; inf if
    the peak touches either end of the grid (reference stats.py:80-102)."""
    try:
        peak = numpy.argmax(power)
        half = 0.5 * power[peak]
        upper = peak + 1
        while power[upper] > half:  # IndexError past the end -> inf
            upper += 1
        lower = peak - 1
        while power[lower] > half:  # walks through negative indices like the reference
            lower -= 1
        return 0.5 * (periods[upper] - periods[lower])
    except Exception:
        return float("inf")


def final_T0_fit(signal, depth, t, y, dy, period, T0_fit_margin, show_progress_bar, verbose,
                 residuals_fn):
    """Mid-transit time of the best (period, duration, depth): chi^2 of the
    depth-scaled template over a grid of trial T0s, first minimum wins.

    Reference stats.py:135-204.  Kept quirk: the weights are 1/flux^2 of the
    flux rolled twice, not 1/dy^2 (the reference overwrites dy with the rolled
    flux, stats.py:191), so `dy` does not influence the result.
    residuals_fn(t, y, period, signal, T0_array, roll) evaluates the loop body (stats.py:178-195)
    for every trial epoch: the device kernel tls_t0_fit (tls_amd.search.t0_fit_residuals).  There
    is no host evaluation in the product; its CPU restatement lives in oracle/tls_oracle.c.
    """
    dur = len(signal)
    scale = C.SIGNAL_DEPTH / (1 - depth)
    signal = 1 - ((1 - signal) / scale)
    n = numpy.size(y)
    if T0_fit_margin == 0:
        points = n
    else:
        points = int(n / (T0_fit_margin * dur))
    if points > n:
        points = n
    T0_array = numpy.linspace(start=numpy.min(t), stop=numpy.min(t) + period, num=points)
    if verbose:
        print("Searching for best T0 for period", format(period, ".5f"), "days")
    roll = int(dur / 2) + 1
    if residuals_fn is None:
        raise RuntimeError("final_T0_fit needs the device evaluation of the trial epochs (no CPU path)")
    total = residuals_fn(t, y, period, signal, T0_array, roll)
    if len(total) == 0 or not (numpy.min(total) < float("inf")):
        return 0
    return T0_array[int(numpy.argmin(total))]  # first strict minimum (stats.py:199-201)


def all_transit_times(T0, t, period):
    """Mid-transit times T0 + k*period inside the time series
    (reference stats.py:244-261)."""
    first = T0 + period if T0 < numpy.min(t) else T0
    end = numpy.min(t) + (numpy.max(t) - numpy.min(t))
    times = [first]
    while times[-1] + period < end:
        times.append(times[-1] + period)
    return times


def calculate_stretch(t, period, transit_times):
    """(time span / period) / number of epochs (reference stats.py:279-291)."""
    return ((numpy.max(t) - numpy.min(t)) / period) / len(transit_times)


def calculate_fill_factor(t):
    """Fraction of cadences present, assuming a constant cadence
    (reference stats.py:294-301)."""
    cadence = numpy.median(numpy.diff(t))
    return (len(t) - 1) / ((numpy.max(t) - numpy.min(t)) / cadence)


def calculate_transit_duration_in_days(t, period, transit_times, duration):
    """Fractional duration -> days, corrected for epochs and gaps
    (reference stats.py:264-276)."""
    raw = duration * calculate_stretch(t, period, transit_times) * period
    return raw * calculate_fill_factor(t)


def model_lightcurve(transit_times, period, t, model_transit_single):
    """Model flux over the whole time series: one template copy per epoch plus one
    before and after, cropped to (min t, max t) (reference stats.py:207-241)."""
    epochs = numpy.concatenate(
        [[transit_times[0] - period], transit_times, [transit_times[-1] + period]])
    samples = int(len(t) / len(transit_times)) * C.OVERSAMPLE_MODEL_LIGHT_CURVE
    xs = numpy.concatenate(
        [numpy.linspace(e - period / 2, e + period / 2, samples) for e in epochs])
    ys = numpy.tile(model_transit_single, len(epochs))
    if numpy.all(numpy.isnan(xs)):
        return None, None
    start = numpy.nanargmax(xs > numpy.min(t))
    stop = numpy.nanargmax(xs > numpy.max(t))
    return ys[start:stop], xs[start:stop]


def _points_between(t, lo, hi):
    return numpy.where(numpy.logical_and(t > lo, t < hi))


def count_stats(t, y, transit_times, transit_duration_in_days):
    """Numbers of points in transit and in equally long windows right before and
    after, over epochs fully inside the data (reference stats.py:304-342)."""
    n_in = n_after = n_before = 0
    d = transit_duration_in_days
    t_first, t_last = numpy.min(t), numpy.max(t)   # (the builtins walk the array element by element)
    for mid in transit_times:
        edges = (mid - 1.5 * d, mid - 0.5 * d, mid + 0.5 * d, mid + 1.5 * d)
        if edges[0] > t_first and edges[3] < t_last:
            n_before += len(y[_points_between(t, edges[0], edges[1])])
            n_in += len(y[_points_between(t, edges[1], edges[2])])
            n_after += len(y[_points_between(t, edges[2], edges[3])])
    return n_in, n_after, n_before


def _mean_and_err(values):
    return numpy.mean(values), numpy.std(values) / numpy.sum(len(values)) ** (0.5)


def intransit_stats(t, y, transit_times, transit_duration_in_days):
    """Per-epoch in-transit flux statistics and the odd/even split
    (reference stats.py:345-416; even = epochs 0, 2, 4, ...)."""
    n_epochs = len(transit_times)
    flux_odd = numpy.array([])
    flux_even = numpy.array([])
    per_transit_count = numpy.zeros([n_epochs])
    transit_depths = numpy.zeros([n_epochs])
    transit_depths_uncertainties = numpy.zeros([n_epochs])
    for i, mid in enumerate(transit_times):
        lo = mid - 0.5 * transit_duration_in_days
        hi = mid + 0.5 * transit_duration_in_days
        if numpy.isnan(lo) or numpy.isnan(hi):
            inside = y[:0]
        else:
            inside = y[_points_between(t, lo, hi)]
        n_inside = numpy.size(inside)
        per_transit_count[i] = n_inside
        if n_inside > 0:
            transit_depths[i] = numpy.mean(inside)
            transit_depths_uncertainties[i] = numpy.std(inside) / numpy.sqrt(n_inside)
        else:
            transit_depths[i] = numpy.nan
            transit_depths_uncertainties[i] = numpy.nan
        if i % 2 == 0:
            flux_even = numpy.append(flux_even, inside)
        else:
            flux_odd = numpy.append(flux_odd, inside)
    mean_odd, err_odd = _mean_and_err(flux_odd) if len(flux_odd) > 0 else (numpy.nan, numpy.nan)
    mean_even, err_even = (_mean_and_err(flux_even) if len(flux_even) > 0
                           else (numpy.nan, numpy.nan))
    return (mean_odd, mean_even, err_odd, err_even, flux_odd, flux_even, per_transit_count,
            transit_depths, transit_depths_uncertainties)


def snr_stats(t, y, period, duration, T0, transit_times, transit_duration_in_days,
              per_transit_count):
    """Per-epoch white-noise and pink-noise SNR (reference stats.py:419-469)."""
    n_epochs = len(transit_times)
    snr_per_transit = numpy.zeros([n_epochs])
    snr_pink_per_transit = numpy.zeros([n_epochs])
    flux_ootr = y[~transit_mask(t, period, 2 * duration, T0)]
    try:
        pinknoise = pink_noise(flux_ootr, int(numpy.mean(per_transit_count)))
    except Exception:
        pinknoise = numpy.nan
    std = numpy.std(flux_ootr) if len(flux_ootr) > 0 else numpy.nan
    for i, mid in enumerate(transit_times):
        lo = mid - 0.5 * transit_duration_in_days
        hi = mid + 0.5 * transit_duration_in_days
        if numpy.isnan(lo) or numpy.isnan(hi):
            inside = y[:0]
        else:
            inside = y[_points_between(t, lo, hi)]
        n_inside = numpy.size(inside)
        mean_flux = numpy.mean(inside) if n_inside > 0 else numpy.nan
        try:
            snr_pink_per_transit[i] = (1 - mean_flux) / pinknoise
            if n_inside > 0 and not numpy.isnan(std):
                snr_per_transit[i] = (1 - mean_flux) / (std / n_inside ** 0.5)
            else:
                snr_per_transit[i] = 0
                snr_pink_per_transit[i] = 0
        except Exception:
            snr_per_transit[i] = 0
            snr_pink_per_transit[i] = 0
    return snr_per_transit, snr_pink_per_transit
