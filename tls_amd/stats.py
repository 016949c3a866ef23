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


def _window_sums(cols):
    """Sum over the window axis of `cols` (a list of equally long vectors: element j of every window),
    in the association numpy's own reduction uses on a contiguous window (pairwise summation: eight
    interleaved partial sums per block of at most 128, blocks halved recursively), so that the result
    equals numpy.sum(window) bit for bit for every window at once."""
    n = len(cols)
    if n < 8:
        total = cols[0].copy()
        for c in cols[1:]:
            total += c
        return total
    if n <= 128:
        r = [cols[j].copy() for j in range(8)]
        full = n - n % 8
        for i in range(8, full, 8):
            for j in range(8):
                r[j] += cols[i + j]
        total = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]))
        for i in range(full, n):
            total += cols[i]
        return total
    half = n // 2
    half -= half % 8
    return _window_sums(cols[:half]) + _window_sums(cols[half:])


def pink_noise(data, width):
    """Mean over all windows of `width` points of std(window)/sqrt(width)
    (reference stats.py:72-77).  The two passes of numpy.std run as `width` shifted vector
    operations over all windows at once, summed in numpy's own order (_window_sums), and the
    running total is accumulated left to right like the reference's loop: bit-identical to it,
    at a tenth of the cost of numpy.std over the short axis of a sliding-window view."""
    data = numpy.ascontiguousarray(data, dtype=float)
    width = int(width)
    n_windows = len(data) - width + 1
    if width < 1 or n_windows < 1:
        raise ValueError("window longer than the data")   # (the reference divides by zero here)
    cols = [data[j:j + n_windows] for j in range(width)]
    mean = _window_sums(cols) / width
    dev = [c - mean for c in cols]
    for v in dev:
        v *= v
    terms = numpy.sqrt(_window_sums(dev) / width) / width ** 0.5
    return numpy.cumsum(terms)[-1] / n_windows


def spectra(chi2, oversampling_factor):
    """SR, power_raw, power, SDE_raw, SDE (reference stats.py:105-132): evaluated on the device
    (tls_amd.search.spectra -> tls_spectra); kept here so that the module reads like the reference's."""
    from . import search as _search
    return _search.spectra(chi2, oversampling_factor)


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


def calculate_transit_duration_in_days(t, period, transit_times, duration, fill_factor=None):
    """Fractional duration -> days, corrected for epochs and gaps
    (reference stats.py:264-276).  fill_factor: calculate_fill_factor(t), if the caller has it already."""
    raw = duration * calculate_stretch(t, period, transit_times) * period
    return raw * (calculate_fill_factor(t) if fill_factor is None else fill_factor)


def model_lightcurve(transit_times, period, t, model_transit_single):
    """Model flux over the whole time series: one template copy per epoch plus one
    before and after, cropped to (min t, max t) (reference stats.py:207-241)."""
    epochs = numpy.concatenate(
        [[transit_times[0] - period], transit_times, [transit_times[-1] + period]])
    samples = int(len(t) / len(transit_times)) * C.OVERSAMPLE_MODEL_LIGHT_CURVE
    # numpy.linspace(e - period / 2, e + period / 2, samples) for every epoch at once: arange * step + start with each
    # epoch's own start, stop and step, the end point set exactly -- the same operations element by element, so the same bits
    # as one linspace call per epoch (13 calls for the 90-day configuration)
    if samples > 1 and not numpy.any(numpy.isnan(epochs)):
        starts, stops = epochs - period / 2, epochs + period / 2
        steps = (stops - starts) / (samples - 1)
        if numpy.all(steps != 0):
            grid = numpy.arange(0, samples, dtype=float)[None, :] * steps[:, None] + starts[:, None]
            grid[:, -1] = stops
            xs = grid.reshape(-1)
        else:
            xs = numpy.concatenate([numpy.linspace(e - period / 2, e + period / 2, samples) for e in epochs])
    else:
        xs = numpy.concatenate([numpy.linspace(e - period / 2, e + period / 2, samples) for e in epochs])
    ys = numpy.tile(model_transit_single, len(epochs))
    if numpy.all(numpy.isnan(xs)):
        return None, None
    start = numpy.nanargmax(xs > numpy.min(t))
    stop = numpy.nanargmax(xs > numpy.max(t))
    return ys[start:stop], xs[start:stop]


def _points_between(t, lo, hi):
    return numpy.where(numpy.logical_and(t > lo, t < hi))


def _is_ascending(t):
    return bool(numpy.all(t[1:] >= t[:-1]))


def _open_interval_slices(t, lo, hi):
    """[start, stop) index ranges of the points with lo < t < hi for arrays of bounds, for
    ASCENDING t (two binary searches per interval instead of two passes over the series).
    NaN bounds give empty ranges, like the reference's comparisons."""
    lo, hi = numpy.asarray(lo, dtype=float), numpy.asarray(hi, dtype=float)
    start = numpy.searchsorted(t, lo, side="right")
    stop = numpy.searchsorted(t, hi, side="left")
    bad = numpy.isnan(lo) | numpy.isnan(hi)
    stop = numpy.where(bad, start, numpy.maximum(stop, start))
    return start, stop


def count_stats(t, y, transit_times, transit_duration_in_days):
    """Numbers of points in transit and in equally long windows right before and
    after, over epochs fully inside the data (reference stats.py:304-342)."""
    d = transit_duration_in_days
    t_first, t_last = numpy.min(t), numpy.max(t)   # (the builtins walk the array element by element)
    if _is_ascending(t):
        mid = numpy.asarray(transit_times, dtype=float)
        edges = [mid - 1.5 * d, mid - 0.5 * d, mid + 0.5 * d, mid + 1.5 * d]
        inside = numpy.logical_and(edges[0] > t_first, edges[3] < t_last)
        counts = []
        for a, b in ((0, 1), (1, 2), (2, 3)):
            start, stop = _open_interval_slices(t, edges[a], edges[b])
            counts.append(int(numpy.sum((stop - start)[inside])))
        return counts[1], counts[2], counts[0]
    n_in = n_after = n_before = 0
    for mid in transit_times:
        edges = (mid - 1.5 * d, mid - 0.5 * d, mid + 0.5 * d, mid + 1.5 * d)
        if edges[0] > t_first and edges[3] < t_last:
            n_before += len(y[_points_between(t, edges[0], edges[1])])
            n_in += len(y[_points_between(t, edges[1], edges[2])])
            n_after += len(y[_points_between(t, edges[2], edges[3])])
    return n_in, n_after, n_before


def _mean_and_err(values):
    return numpy.mean(values), numpy.std(values) / numpy.sum(len(values)) ** (0.5)


def _intransit_fluxes(t, y, transit_times, transit_duration_in_days):
    """The in-transit flux of every epoch (mid - d/2 < t < mid + d/2), as a list of arrays."""
    mid = numpy.asarray(transit_times, dtype=float)
    lo = mid - 0.5 * transit_duration_in_days
    hi = mid + 0.5 * transit_duration_in_days
    if _is_ascending(t):
        start, stop = _open_interval_slices(t, lo, hi)
        return [y[a:b] for a, b in zip(start.tolist(), stop.tolist())]
    out = []
    for a, b in zip(lo, hi):
        out.append(y[:0] if (numpy.isnan(a) or numpy.isnan(b)) else y[_points_between(t, a, b)])
    return out


_EXACT_SEGMENTS = 512   # epochs up to which the per-epoch statistics use the reference's numpy calls one by one


def _segment_means_and_stds(chunks):
    """(size, mean, population std) of every chunk, NaN for empty chunks: numpy.mean / numpy.std per chunk (bit-equal to
    the reference) up to _EXACT_SEGMENTS chunks, one segmented two-pass reduction (1-2 ulp from it) beyond."""
    sizes = numpy.array([len(c) for c in chunks], dtype=numpy.int64)
    means = numpy.full(len(chunks), numpy.nan)
    stds = numpy.full(len(chunks), numpy.nan)
    filled = numpy.nonzero(sizes)[0]
    if 0 < len(filled) <= _EXACT_SEGMENTS:
        # the reference's own calls, epoch by epoch (stats.py:345-469: numpy.mean / numpy.std sum pairwise from eight
        # elements on): the same bits.  The segmented form below adds left to right and differs in the last 1-2 ulp
        # (4e-16 relative) on three light curves in four; it is kept for light curves with thousands of epochs.
        for i in filled:
            means[i] = numpy.mean(chunks[i])
            stds[i] = numpy.std(chunks[i])
    elif len(filled):
        flat = numpy.concatenate([chunks[i] for i in filled])
        cnt = sizes[filled]
        starts = numpy.concatenate([[0], numpy.cumsum(cnt)[:-1]])
        m = numpy.add.reduceat(flat, starts) / cnt
        dev = flat - numpy.repeat(m, cnt)
        means[filled] = m
        stds[filled] = numpy.sqrt(numpy.add.reduceat(dev * dev, starts) / cnt)
    return sizes, means, stds


def intransit_stats(t, y, transit_times, transit_duration_in_days, chunks=None):
    """Per-epoch in-transit flux statistics and the odd/even split
    (reference stats.py:345-416; even = epochs 0, 2, 4, ...)."""
    if chunks is None:
        chunks = _intransit_fluxes(t, y, transit_times, transit_duration_in_days)
    sizes, transit_depths, stds = _segment_means_and_stds(chunks)
    per_transit_count = sizes.astype(float)
    with numpy.errstate(invalid="ignore", divide="ignore"):
        transit_depths_uncertainties = stds / numpy.sqrt(per_transit_count)
    empty = numpy.zeros(0)
    flux_even = numpy.concatenate([empty] + chunks[0::2])
    flux_odd = numpy.concatenate([empty] + chunks[1::2])
    mean_odd, err_odd = _mean_and_err(flux_odd) if len(flux_odd) > 0 else (numpy.nan, numpy.nan)
    mean_even, err_even = (_mean_and_err(flux_even) if len(flux_even) > 0
                           else (numpy.nan, numpy.nan))
    return (mean_odd, mean_even, err_odd, err_even, flux_odd, flux_even, per_transit_count,
            transit_depths, transit_depths_uncertainties)


def snr_stats(t, y, period, duration, T0, transit_times, transit_duration_in_days,
              per_transit_count, chunks=None, flux_ootr=None, mean_flux=None, std_ootr=None, pink_noise_fn=None):
    """Per-epoch white-noise and pink-noise SNR (reference stats.py:419-469).  `mean_flux` (the per-epoch means
    intransit_stats has just formed from the same chunks) and `std_ootr` (numpy.std(flux_ootr)) may be handed in;
    `pink_noise_fn(data, width)`: power() hands in the device form (search.pink_noise, the same bits)."""
    if flux_ootr is None:
        flux_ootr = y[~transit_mask(t, period, 2 * duration, T0)]
    try:
        pinknoise = (pink_noise_fn or pink_noise)(flux_ootr, int(numpy.mean(per_transit_count)))
    except Exception:
        pinknoise = numpy.nan
    if std_ootr is not None and len(flux_ootr) > 0:
        std = std_ootr
    else:
        std = numpy.std(flux_ootr) if len(flux_ootr) > 0 else numpy.nan
    if mean_flux is not None:
        sizes = numpy.asarray(per_transit_count).astype(numpy.int64)
    else:
        if chunks is None:
            chunks = _intransit_fluxes(t, y, transit_times, transit_duration_in_days)
        sizes, mean_flux, _ = _segment_means_and_stds(chunks)
    with numpy.errstate(invalid="ignore", divide="ignore"):
        snr_pink = (1 - mean_flux) / pinknoise
        snr_white = (1 - mean_flux) / (std / numpy.sqrt(sizes.astype(float)))
    usable = numpy.logical_and(sizes > 0, not numpy.isnan(std))
    snr_per_transit = numpy.where(usable, snr_white, 0.0)
    snr_pink_per_transit = numpy.where(usable, snr_pink, 0.0)
    return snr_per_transit, snr_pink_per_transit
