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

    if numpy.max(t) - numpy.min(t) <= 0:
        raise ValueError("Time duration must positive")
    if numpy.size(y) < 3 or numpy.size(t) < 3:
        raise ValueError("Too few values in data set")
    flux_mean = numpy.mean(y)
    if flux_mean > 1.01 or flux_mean < 0.99:
        warnings.warn("Warning: The mean flux should be normalized to 1"
                      + ", but it was found to be " + str(flux_mean))
    if numpy.min(y) < 0:
        raise ValueError("Flux values must be positive")
    if numpy.max(y) >= float("inf"):
        raise ValueError("Flux values must be finite")

    if dy is None:
        dy = numpy.full(len(y), numpy.std(y))
    if numpy.size(t) != numpy.size(y) or numpy.size(t) != numpy.size(dy):
        raise ValueError("Arrays (t, y, dy) must be of the same dimensions")
    if t.ndim != 1:
        raise ValueError("Inputs (t, y, dy) must be 1-dimensional")
    return t, y, dy


def _not_positive_finite(value):
    return value <= 0 or value >= float("inf")


# keyword -> default, for the parameters that are simply "the caller's value or the default"
_PLAIN_PARAMETERS = (
    ("show_progress_bar", True), ("transit_depth_min", C.TRANSIT_DEPTH_MIN),
    ("R_star", C.R_STAR), ("M_star", C.M_STAR), ("oversampling_factor", C.OVERSAMPLING_FACTOR),
    ("period_max", float("inf")), ("period_min", 0), ("n_transits_min", C.N_TRANSITS_MIN),
    ("R_star_min", C.R_STAR_MIN), ("R_star_max", C.R_STAR_MAX),
    ("M_star_min", C.M_STAR_MIN), ("M_star_max", C.M_STAR_MAX),
    ("duration_grid_step", C.DURATION_GRID_STEP),
    ("per", C.DEFAULT_PERIOD), ("rp", C.DEFAULT_RP), ("a", C.DEFAULT_A),
    ("T0_fit_margin", C.T0_FIT_MARGIN), ("ecc", C.DEFAULT_ECC), ("w", C.DEFAULT_W),
    ("u", C.DEFAULT_U), ("limb_dark", C.DEFAULT_LIMB_DARK),
)

# range checks in the reference's order (the first violated one is the one reported,
# validate.py:122-175): (violated?, message)
_RANGE_RULES = (
    (lambda s: _not_positive_finite(s.R_star), "R_star must be positive"),
    (lambda s: s.R_star_min > s.R_star, "R_star_min <= R_star is required"),
    (lambda s: _not_positive_finite(s.R_star_min), "R_star_min must be positive"),
    (lambda s: s.R_star_max < s.R_star, "R_star_max >= R_star is required"),
    (lambda s: _not_positive_finite(s.R_star_max), "R_star_max must be positive"),
    (lambda s: _not_positive_finite(s.M_star), "M_star must be positive"),
    (lambda s: s.M_star_min > s.M_star, "M_star_min <= M_star is required"),
    (lambda s: _not_positive_finite(s.M_star_min), "M_star_min must be positive"),
    (lambda s: s.M_star_max < s.M_star, "M_star_max >= M_star required"),
    (lambda s: _not_positive_finite(s.M_star_max), "M_star_max must be positive"),
    (lambda s: s.period_min < 0, "period_min >= 0 required"),
    (lambda s: s.period_min >= s.period_max, "period_min < period_max required"),
    (lambda s: not isinstance(s.n_transits_min, int), "n_transits_min must be an integer value"),
    (lambda s: s.n_transits_min < 1, "n_transits_min must be an integer value >= 1"),
    (lambda s: not isinstance(s.use_threads, int) or s.use_threads < 1,
     "use_threads must be an integer value >= 1"),
)


def _apply_template_preset(self):
    """transit_template selects the shape of the template planet (validate.py:101-119)."""
    name = self.transit_template
    if name == "default":
        # the default preset overrides per/rp/a/inc given as kwargs
        self.per, self.rp, self.a, self.inc = (
            C.DEFAULT_PERIOD, C.DEFAULT_RP, C.DEFAULT_A, C.DEFAULT_INC)
    elif name == "grazing":
        self.b = C.GRAZING_B
        self.inc = impact_to_inclination(b=self.b, semimajor_axis=self.a)
    elif name == "box":
        self.per, self.rp, self.a = C.BOX_PERIOD, C.BOX_RP, C.BOX_A
        self.b, self.inc = C.BOX_B, C.BOX_INC
        self.u, self.limb_dark = C.BOX_U, C.BOX_LIMB_DARK
    else:
        raise ValueError('Unknown transit_template. Known values: \
            "default", "grazing", "box"')


def validate_args(self, kwargs):
    """Fill `self` with the search parameters: kwargs or defaults, then the
    template preset, then range checks (reference validate.py:49-181)."""
    self.verbose = kwargs.get("verbose", True)
    for key in kwargs:
        if key not in C.VALID_PARAMETERS and key not in C.EXTRA_PARAMETERS:
            warnings.warn("Ignoring unknown parameter: " + str(key))

    for name, default in _PLAIN_PARAMETERS:
        setattr(self, name, kwargs.get(name, default))
    self.use_threads = kwargs.get("use_threads", multiprocessing.cpu_count())
    if "b" in kwargs:  # an impact parameter overrules the inclination
        self.b = kwargs["b"]
        self.inc = impact_to_inclination(b=self.b, semimajor_axis=self.a)
    else:
        self.inc = kwargs.get("inc", C.DEFAULT_INC)
    self.transit_template = kwargs.get("transit_template", "default")
    _apply_template_preset(self)

    for violated, message in _RANGE_RULES:
        if violated(self):
            raise ValueError(message)

    # the T0 stride is clamped to [0, 10 %] of the transit duration (validate.py:177-180)
    self.T0_fit_margin = min(max(self.T0_fit_margin, 0), 0.1)
    return self, kwargs
