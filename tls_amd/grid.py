"""Trial grids of the search: transit-duration bounds, duration grid, period grid.

Host-side numpy. Behaviour (lengths, end points, warnings, the fall-back for
tiny grids) follows the reference's transitleastsquares/grid.py:9-156, because
the grids define the work list the HIP kernels run over and their exact
lengths/end points are pinned by the reference's own tests
(tests/test_period_grid.py:8-50, tests/test_duration_grid.py:7-18).
"""
import warnings

import numpy

from . import constants as C


def T14(R_s, M_s, P, upper_limit=C.FRACTIONAL_TRANSIT_DURATION_MAX, small=False):
    """Longest transit duration T14 as a FRACTION of the period P [days].

    R_s, M_s in solar units. small=True drops the planet radius (point planet);
    otherwise a 2 R_jup planet is assumed.  Capped at upper_limit.
    Reference: grid.py:9-32 (same operation order, so the same bits in CPython).
    """
    P = P * C.SECONDS_PER_DAY
    R_s = C.R_sun * R_s
    M_s = C.M_sun * M_s
    chord = ((4 * P) / (numpy.pi * C.G * M_s)) ** (1 / 3)
    if small:
        T14max = R_s * chord
    else:
        T14max = (R_s + 2 * C.R_jup) * chord
    result = T14max / P
    if result > upper_limit:
        result = upper_limit
    return result


def duration_grid(periods, shortest, log_step=C.DURATION_GRID_STEP):
    """Geometric grid of fractional durations between the shortest plausible
    transit at the longest period and the longest one at the shortest period.
    `shortest` is accepted and unused, as in the reference (grid.py:35-56)."""
    longest = T14(R_s=C.R_STAR_MAX, M_s=C.M_STAR_MAX, P=min(periods), small=False)
    current = T14(R_s=C.R_STAR_MIN, M_s=C.M_STAR_MIN, P=max(periods), small=True)
    durations = [current]
    while current * log_step < longest:
        current = current * log_step
        durations.append(current)
    durations.append(longest)  # end point, not on the geometric ladder
    return durations


def _clamp_star(value, low, high, reset_low, name):
    # the radius lower clamp resets to 0.1 although the message says 0.01
    # (grid.py:71-78) -- kept, grids must be identical.
    if value < low:
        warnings.warn("Warning: %s was set to %s for period_grid (was unphysical: %s)"
                      % (name, str(low), str(value)))
        return reset_low
    if value > high:
        warnings.warn("Warning: %s was set to %s for period_grid (was unphysical: %s)"
                      % (name, str(high), str(value)))
        return high
    return value


def period_grid(R_star, M_star, time_span, period_min=0, period_max=float("inf"),
                oversampling_factor=C.OVERSAMPLING_FACTOR,
                n_transits_min=C.N_TRANSITS_MIN):
    """Optimal period sampling of Ofir (2014, A&A 561, A138): uniform in
    frequency**(1/3).  Returns periods in days, DESCENDING.
    Reference: grid.py:59-156."""
    R_star = _clamp_star(R_star, 0.01, 10000, 0.1, "R_star")
    M_star = _clamp_star(M_star, 0.01, 1000, 0.01, "M_star")

    R_star = R_star * C.R_sun
    M_star = M_star * C.M_sun
    time_span = time_span * C.SECONDS_PER_DAY

    f_min = n_transits_min / time_span
    f_max = 1.0 / (2 * numpy.pi) * numpy.sqrt(C.G * M_star / (3 * R_star) ** 3)

    # Ofir eq. 5-7
    A = ((2 * numpy.pi) ** (2.0 / 3) / numpy.pi * R_star
         / (C.G * M_star) ** (1.0 / 3) / (time_span * oversampling_factor))
    C0 = f_min ** (1.0 / 3) - A / 3.0
    N_opt = (f_max ** (1.0 / 3) - f_min ** (1.0 / 3) + A / 3) * 3 / A

    X = numpy.arange(N_opt) + 1  # N_opt is a float: arange rounds the count up
    periods = 1 / (A / 3 * X + C0) ** 3 / C.SECONDS_PER_DAY
    keep = numpy.where(numpy.logical_and(periods > period_min, periods <= period_max))
    n_keep = numpy.size(periods[keep])

    if n_keep > 10 ** 6:
        warnings.warn("period_grid generates a very large grid (" + str(n_keep)
                      + "). Recommend to check physical plausibility for stellar mass,"
                      " radius, and time series duration.")
    if n_keep < C.MINIMUM_PERIOD_GRID_SIZE:
        if time_span < 5 * C.SECONDS_PER_DAY:
            time_span = 5 * C.SECONDS_PER_DAY
        warnings.warn("period_grid defaults to R_star=1 and M_star=1 as given density"
                      " yielded grid with too few values")
        return period_grid(R_star=1, M_star=1, time_span=time_span / C.SECONDS_PER_DAY)
    return periods[keep]
