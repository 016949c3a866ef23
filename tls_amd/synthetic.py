"""Synthetic light curves of the benchmark configurations (SURVEY.md section 8d,
BASELINE.json `configs`): Tutorial-01 planet injected into white noise.

Generator: numpy.random.seed(seed) (legacy RandomState stream), t = linspace(3.14,
3.14 + span, int(span * cadence_per_day)), planet t0 = 3.14, per = 10.123 d,
rp = 6371/696342, a = 19, inc = 90, quadratic limb darkening [0.4, 0.4]
(reference tutorials/01 Quick start with synthetic data.ipynb:23-48).
"""
import numpy

from . import transit_model

TIME_START = 3.14

CONFIGS = {
    # name: (span [d], cadences per day, noise sigma, power() kwargs)
    "tutorial01": (100.0, 48, 50e-6, {}),
    "k2_90d": (90.0, 48, 50e-6, {}),
    "kepler_4yr": (1461.0, 48, 50e-6, {"period_min": 0.5, "period_max": 400}),
    "tess_27d": (27.0, 720, 200e-6, {}),
    # (not a BASELINE configuration: a series between the four-slot kernel's 5120 points and the LDS-resident limit -- two
    # Kepler quarters, two TESS sectors at 10 min -- for the 512-thread shape of that kernel; tests and probes)
    "lc_150d": (150.0, 48, 50e-6, {}),
}


def light_curve(span_days, cadence_per_day, sigma, seed=0, per=10.123, rp=6371 / 696342,
                a=19, inc=90, u=(0.4, 0.4)):
    """(t, flux) with the planet injected and Gaussian noise of std `sigma`."""
    numpy.random.seed(seed)
    n = int(span_days * cadence_per_day)
    t = numpy.linspace(TIME_START, TIME_START + span_days, n)
    signal = transit_model.light_curve(t, TIME_START, per, rp, a, inc, 0, 90, list(u), "quadratic")
    flux = signal + numpy.random.normal(0, sigma, n)
    return t, flux


def config(name, seed=0, sigma=None):
    """(t, flux, power_kwargs) of a named benchmark configuration."""
    span, cadence, sig, kwargs = CONFIGS[name]
    t, flux = light_curve(span, cadence, sig if sigma is None else sigma, seed=seed)
    return t, flux, dict(kwargs)


from .planning import search_inputs  # noqa: E402,F401  (benchmarks and tests reach it through this module too)
