"""Input and keyword validation of the public API.

Defines the numeric inputs the kernels see (dy normalisation, dy = std(y) when
absent) and keeps the reference's exception types and message texts
(validate.py:9-181) so that callers' error handling keeps working.
"""
import multiprocessing
import warnings

import numpy

from . import constants as C
from .helpers import cleaned_array, impact_to_inclination


def validate_inputs(t, y, dy):
    """Clean (t, y[, dy]); normalise dy by its mean, or use std(y) when no dy is
    given (reference validate.py:9-46)."""
    if dy is None:
        t, y = cleaned_array(t, y)
    else:
        t, y, dy = cleaned_array(t, y, dy)
        dy = dy / numpy.mean(dy)  # weights only: scale is irrelevant

    if max(t) - min(t) <= 0:
        raise ValueError("Time duration must positive")
    if numpy.size(y) < 3 or numpy.size(t) < 3:
        raise ValueError("Too few values in data set")
    flux_mean = numpy.mean(y)
    if flux_mean > 1.01 or flux_mean < 0.99:
        warnings.warn("Warning: The mean flux should be normalized to 1"
                      + ", but it was found to be " + str(flux_mean))
    if min(y) < 0:
        raise ValueError("Flux values must be positive")
    if max(y) >= float("inf"):
        raise ValueError("Flux values must be finite")

    if dy is None:
        dy = numpy.full(len(y), numpy.std(y))
    if numpy.size(t) != numpy.size(y) or numpy.size(t) != numpy.size(dy):
        raise ValueError("Arrays (t, y, dy) must be of the same dimensions")
    if t.ndim != 1:
        raise ValueError("Inputs (t, y, dy) must be 1-dimensional")
    return t, y, dy


def _require_positive_finite(value, message):
    if value <= 0 or value >= float("inf"):
        raise ValueError(message)


def validate_args(self, kwargs):
    """Fill `self` with the search parameters: kwargs or defaults, then the
    template preset, then range checks (reference validate.py:49-181)."""
    self.verbose = kwargs.get("verbose", True)
    for key in kwargs:
        if key not in C.VALID_PARAMETERS and key not in C.EXTRA_PARAMETERS:
            warnings.warn("Ignoring unknown parameter: " + str(key))

    get = kwargs.get
    self.show_progress_bar = get("show_progress_bar", True)
    self.transit_depth_min = get("transit_depth_min", C.TRANSIT_DEPTH_MIN)
    self.R_star = get("R_star", C.R_STAR)
    self.M_star = get("M_star", C.M_STAR)
    self.oversampling_factor = get("oversampling_factor", C.OVERSAMPLING_FACTOR)
    self.period_max = get("period_max", float("inf"))
    self.period_min = get("period_min", 0)
    self.n_transits_min = get("n_transits_min", C.N_TRANSITS_MIN)
    self.R_star_min = get("R_star_min", C.R_STAR_MIN)
    self.R_star_max = get("R_star_max", C.R_STAR_MAX)
    self.M_star_min = get("M_star_min", C.M_STAR_MIN)
    self.M_star_max = get("M_star_max", C.M_STAR_MAX)
    self.duration_grid_step = get("duration_grid_step", C.DURATION_GRID_STEP)
    self.use_threads = get("use_threads", multiprocessing.cpu_count())
    self.per = get("per", C.DEFAULT_PERIOD)
    self.rp = get("rp", C.DEFAULT_RP)
    self.a = get("a", C.DEFAULT_A)
    self.T0_fit_margin = get("T0_fit_margin", C.T0_FIT_MARGIN)

    if "b" in kwargs:  # an impact parameter overrules the inclination
        self.b = get("b")
        self.inc = impact_to_inclination(b=self.b, semimajor_axis=self.a)
    else:
        self.inc = get("inc", C.DEFAULT_INC)
    self.ecc = get("ecc", C.DEFAULT_ECC)
    self.w = get("w", C.DEFAULT_W)
    self.u = get("u", C.DEFAULT_U)
    self.limb_dark = get("limb_dark", C.DEFAULT_LIMB_DARK)

    self.transit_template = get("transit_template", "default")
    if self.transit_template == "default":
        # the default preset overrides per/rp/a/inc given as kwargs (validate.py:101-105)
        self.per, self.rp, self.a, self.inc = (
            C.DEFAULT_PERIOD, C.DEFAULT_RP, C.DEFAULT_A, C.DEFAULT_INC)
    elif self.transit_template == "grazing":
        self.b = C.GRAZING_B
        self.inc = impact_to_inclination(b=self.b, semimajor_axis=self.a)
    elif self.transit_template == "box":
        self.per, self.rp, self.a = C.BOX_PERIOD, C.BOX_RP, C.BOX_A
        self.b, self.inc = C.BOX_B, C.BOX_INC
        self.u, self.limb_dark = C.BOX_U, C.BOX_LIMB_DARK
    else:
        raise ValueError('Unknown transit_template. Known values: \
            "default", "grazing", "box"')

    _require_positive_finite(self.R_star, "R_star must be positive")
    if self.R_star_min > self.R_star:
        raise ValueError("R_star_min <= R_star is required")
    _require_positive_finite(self.R_star_min, "R_star_min must be positive")
    if self.R_star_max < self.R_star:
        raise ValueError("R_star_max >= R_star is required")
    _require_positive_finite(self.R_star_max, "R_star_max must be positive")

    _require_positive_finite(self.M_star, "M_star must be positive")
    if self.M_star_min > self.M_star:
        raise ValueError("M_star_min <= M_star is required")
    _require_positive_finite(self.M_star_min, "M_star_min must be positive")
    if self.M_star_max < self.M_star:
        raise ValueError("M_star_max >= M_star required")
    _require_positive_finite(self.M_star_max, "M_star_max must be positive")

    if self.period_min < 0:
        raise ValueError("period_min >= 0 required")
    if self.period_min >= self.period_max:
        raise ValueError("period_min < period_max required")
    if not isinstance(self.n_transits_min, int):
        raise ValueError("n_transits_min must be an integer value")
    if self.n_transits_min < 1:
        raise ValueError("n_transits_min must be an integer value >= 1")
    if not isinstance(self.use_threads, int) or self.use_threads < 1:
        raise ValueError("use_threads must be an integer value >= 1")

    # the T0 stride is clamped to [0, 10 %] of the transit duration (validate.py:177-180)
    if self.T0_fit_margin < 0:
        self.T0_fit_margin = 0
    elif self.T0_fit_margin > 0.1:
        self.T0_fit_margin = 0.1
    return self, kwargs
