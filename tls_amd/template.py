"""Transit template table: one limb-darkened transit shape, resampled to every
trial duration (in samples) the search uses.

This is the input of the HIP search path, built once per power() call on the
host.  It reproduces the reference's table exactly (transit.py:8-160): same
supersampled curve, same asymmetric in-transit slice, same resampling and the
same `overshoot` factor, because chi^2 parity at the 1e-6 level depends on it.

Two representations are returned/used:
* the reference's own (`lc_cache_overview` structured array + `lc_arr` object
  array of ragged rows) so host code reads like the reference's, and
* `TemplateTable`, the flat layout the C ABI takes (values/offset/width/overshoot).
"""
import numpy

from . import constants as C
from . import transit_model
from .interp import interp1d


_CURVES = {}  # shape parameters -> (t, flux, first in-transit sample): the model is the costly part
_SHAPES = {}  # (samples, shape parameters) -> the resampled in-transit shape (power() asks for the same few per call)


def _supersampled_curve(per, rp, a, inc, ecc, w, u, limb_dark):
    """The template planet's light curve over +-0.5 d around mid-transit (transit.py:11-25).  It
    depends on the shape parameters only, and power() asks for it several times per call."""
    key = (float(per), float(rp), float(a), float(inc), float(ecc), float(w),
           tuple(float(x) for x in u), str(limb_dark))
    hit = _CURVES.get(key)
    if hit is None:
        half = 0.5
        t = numpy.linspace(-half, half, C.SUPERSAMPLE_SIZE)
        flux = transit_model.light_curve(t, 0, per, rp, a, inc, ecc, w, u, limb_dark)
        first = int(numpy.argmax(flux < 1))  # first in-transit sample
        if len(_CURVES) >= 8:
            _CURVES.pop(next(iter(_CURVES)))
        hit = _CURVES[key] = (t, flux, first)
    return hit


def reference_transit(samples, per, rp, a, inc, ecc, w, u, limb_dark):
    """In-transit part of the template planet's light curve, resampled to
    `samples` points and rescaled to depth 1 (0 = bottom, 1 = out of transit).
    Reference transit.py:8-42."""
    key = (int(samples), float(per), float(rp), float(a), float(inc), float(ecc), float(w),
           tuple(float(x) for x in u), str(limb_dark))
    hit = _SHAPES.get(key)
    if hit is None:
        t, flux, first = _supersampled_curve(per, rp, a, inc, ecc, w, u, limb_dark)
        # the slice is one sample longer on the egress side (transit.py:29-30)
        in_flux = flux[first: -first + 1]
        in_time = t[first: -first + 1]
        x_new = numpy.linspace(t[first], t[-first - 1], samples)
        down = interp1d(x_new, in_time)(in_flux)
        lo = numpy.min(down)
        hit = (lo - down) / (lo - 1)
        hit.setflags(write=False)   # shared between calls
        if len(_SHAPES) >= 32:
            _SHAPES.pop(next(iter(_SHAPES)))
        _SHAPES[key] = hit
    return hit


def fractional_transit(duration, maxwidth, depth, samples, per, rp, a, inc, ecc, w, u,
                       limb_dark, cached_reference_transit=None):
    """The reference transit squeezed to duration/maxwidth of a `samples`-long
    window (padded with ones) and scaled to `depth`.  Reference transit.py:45-95."""
    if cached_reference_transit is None:
        shape = reference_transit(samples, per, rp, a, inc, ecc, w, u, limb_dark)
    else:
        shape = cached_reference_transit

    grid = numpy.linspace(-0.5, 0.5, samples)
    occupied = int((duration / maxwidth) * samples)
    squeezed = interp1d(numpy.linspace(-0.5, 0.5, occupied), grid)(shape)

    pad = numpy.ones(int((samples - occupied) * 0.5))
    result = numpy.concatenate([pad, squeezed, pad])
    if numpy.size(result) < samples:  # odd remainder goes to the right
        result = numpy.append(result, numpy.ones(1))
    return 1 - ((1 - result) * depth)


class TemplateTable(object):
    """Flat, C-ABI-ready view of the template rows.

    values    f8[sum(len(row))]  all rows back to back
    offset    i8[rows]           start of row r in `values`
    length    i8[rows]           len(row r) (== width in practice, see a11 in SURVEY.md)
    width     i8[rows]           trial duration of row r in samples
    overshoot f8[rows]           1 / (2 - mean(row)/min(row))
    duration  f8[rows]           fractional duration of row r
    """

    def __init__(self, overview, rows):
        self.n_rows = len(rows)
        self.length = numpy.array([len(r) for r in rows], dtype=numpy.int64)
        self.offset = numpy.zeros(self.n_rows, dtype=numpy.int64)
        if self.n_rows > 1:
            self.offset[1:] = numpy.cumsum(self.length)[:-1]
        self.values = (numpy.concatenate([numpy.asarray(r, dtype=numpy.float64) for r in rows])
                       if self.n_rows else numpy.zeros(0))
        self.values = numpy.ascontiguousarray(self.values, dtype=numpy.float64)
        self.width = numpy.ascontiguousarray(overview["width_in_samples"], dtype=numpy.int64)
        self.overshoot = numpy.ascontiguousarray(overview["overshoot"], dtype=numpy.float64)
        self.duration = numpy.ascontiguousarray(overview["duration"], dtype=numpy.float64)

    def row(self, r):
        return self.values[self.offset[r]: self.offset[r] + self.length[r]]


def get_cache(durations, maxwidth_in_samples, per, rp, a, inc, ecc, w, u, limb_dark,
              verbose=True):
    """Template rows for every fractional duration in `durations`.
    Returns (lc_cache_overview, lc_arr) like the reference's transit.py:98-160."""
    if verbose:
        print("Creating model cache for", str(len(durations)), "durations")
    n_rows = numpy.size(durations)
    overview = numpy.zeros(
        n_rows, dtype=[("duration", "f8"), ("width_in_samples", "i8"), ("overshoot", "f8")])
    shape = reference_transit(maxwidth_in_samples, per, rp, a, inc, ecc, w, u, limb_dark)
    longest = numpy.max(durations)
    rows = []
    for r, duration in enumerate(durations):
        scaled = fractional_transit(
            duration=duration, maxwidth=longest, depth=C.SIGNAL_DEPTH,
            samples=maxwidth_in_samples, per=per, rp=rp, a=a, inc=inc, ecc=ecc, w=w,
            u=u, limb_dark=limb_dark, cached_reference_transit=shape)
        overview["duration"][r] = duration
        overview["width_in_samples"][r] = int((duration / longest) * maxwidth_in_samples)
        # keep the part that is measurably below 1 (transit.py:143-149)
        below = numpy.where(scaled < (1 - C.NUMERICAL_STABILITY_CUTOFF))
        signal = scaled[numpy.min(below): numpy.max(below) + 1]
        rows.append(signal)
        overview["overshoot"][r] = 1 / (2 - numpy.mean(signal) / numpy.min(signal))
    lc_arr = numpy.empty(n_rows, dtype=object)
    for r, signal in enumerate(rows):
        lc_arr[r] = signal
    return overview, lc_arr
