"""Piecewise-linear resampling used by the template builder and resample().

The reference (interpolation.py:7-58) does this with a numba interpolation
search; here it is one vectorised numpy bracket lookup.  The bracket rule is
the same -- index j with x[j] <= z < x[j+1], clamped to [0, n-2] so points
outside the grid extrapolate from the end segments -- and the blend is the same
expression, so results are bit-identical on monotonic grids.
"""
import numpy


class interp1d(object):
    """interp1d(x_new, x)(y): values of the polyline (x, y) at x_new."""

    def __init__(self, x_new, x):
        x = numpy.asarray(x, dtype=float)
        x_new = numpy.asarray(x_new, dtype=float)
        n = x.size
        assert n > 1
        j = numpy.searchsorted(x, x_new, side="right") - 1
        j = numpy.clip(j, 0, n - 2).astype(numpy.int64)
        self._index = j
        self._theta = (x_new - x[j]) / (x[j + 1] - x[j])

    def __call__(self, y):
        y = numpy.asarray(y, dtype=float)
        j, theta = self._index, self._theta
        return (1 - theta) * y[j] + theta * y[j + 1]
