"""Small array utilities that are part of the reference's public surface
(helpers.py:8-113, core.py:9-12): cleaning, resampling, masks, running
statistics, phase folding.  Host-side numpy."""
import numpy

from .interp import interp1d


def fold(time, period, T0):
    """Phase in [0, 1) of each time stamp for a given period and epoch
    (reference core.py:9-12)."""
    x = (time - T0) / period
    return x - numpy.floor(x)


def resample(time, flux, factor):
    """Linear re-binning of a light curve onto len(flux)/factor equidistant
    points (reference helpers.py:8-15)."""
    n_out = int(len(flux) / factor)
    time_resampled = numpy.linspace(numpy.min(time), numpy.max(time), n_out)
    flux_resampled = interp1d(time_resampled, time)(flux)
    return time_resampled, flux_resampled


def _usable(v):
    # finite, strictly positive, not None (reference helpers.py:22-28)
    if v is None:
        return False
    try:
        return bool(v > 0) and bool(v < numpy.inf)  # NaN fails both comparisons
    except TypeError:
        return False


def cleaned_array(t, y, dy=None):
    """Drop every cadence where t, y (or dy) is None, NaN, infinite or not
    positive; accepts object arrays and masked arrays; returns float arrays
    (reference helpers.py:18-61)."""
    cols = (t, y) if dy is None else (t, y, dy)
    keep = [i for i in range(len(y)) if all(_usable(c[i]) for c in cols)]
    out = tuple(numpy.array([c[i] for i in keep], dtype=float) for c in cols)
    return out


def transit_mask(t, period, duration, T0):
    """True for cadences within duration/2 of a transit centre
    (reference helpers.py:64-67)."""
    return numpy.abs((t - T0 + 0.5 * period) % period - 0.5 * period) < 0.5 * duration


def running_mean(data, width_signal):
    """Mean over a sliding window of width_signal samples, via a prefix sum
    (reference helpers.py:70-73).  Length len(data) - width + 1."""
    cumsum = numpy.cumsum(numpy.insert(data, 0, 0))
    return (cumsum[width_signal:] - cumsum[:-width_signal]) / float(width_signal)


def _pad_to(values, n):
    # repeat the end values so the result has n entries (helpers.py:84-90,99-107)
    missing = n - len(values)
    front = int(missing * 0.5)
    return numpy.concatenate(
        [numpy.full(front, values[0]), values, numpy.full(missing - front, values[-1])])


def running_mean_equal_length(data, width_signal):
    """running_mean padded to len(data) (reference helpers.py:76-90)."""
    return _pad_to(running_mean(data, width_signal), len(data))


def running_median(data, kernel):
    """Sliding median of width `kernel`, padded to len(data)
    (reference helpers.py:93-108).  For odd kernels the median is an element of the
    window, so scipy's rank filter (when scipy is installed; it is optional) returns the
    identical values ~40x faster; otherwise, and for even kernels (mean of the two middle
    elements), the numpy formulation.  A non-integer kernel counts as the reference's
    numpy.arange(kernel) does: rounded up."""
    data = numpy.asarray(data, dtype=float)
    kernel = int(numpy.ceil(kernel))
    med = None
    if kernel % 2 == 1 and 1 < kernel <= len(data):
        try:
            from scipy import ndimage
            half = kernel // 2
            med = ndimage.median_filter(data, size=kernel, mode="nearest")[half: len(data) - half]
        except ImportError:
            med = None
    if med is None:
        windows = numpy.lib.stride_tricks.sliding_window_view(data, kernel)
        med = numpy.median(windows, axis=1)
    return _pad_to(med, len(data))


def impact_to_inclination(b, semimajor_axis):
    """Impact parameter -> inclination in degrees (reference helpers.py:111-113)."""
    return numpy.degrees(numpy.arccos(b / semimajor_axis))
