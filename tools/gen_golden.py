"""Generate tests/golden/*.npz by running the UNMODIFIED reference (build container only).

The reference package at /root/reference is imported through tools/ref_shim.py
(numba -> identity decorators, batman -> this repo's Mandel & Agol restatement).
Every array written here is an INPUT or an OUTPUT of reference code:

  search_*.npz   inputs of transitleastsquares.core.search_period (t, y, dy, periods,
                 the template table as produced by transitleastsquares.transit.get_cache,
                 scalars) and its outputs (chi2, row, depth) for each listed period.
  power_*.npz    inputs (t, y, dy, kwargs) and the scalar/array results of
                 transitleastsquares.main.transitleastsquares(...).power(...).
  grids.npz      period_grid / duration_grid / template-table outputs for a few argument sets.
  k2_*.npz       the two K2 light curves the reference's tests hold as data fixtures
                 (tests/EPIC201367065.csv, tests/EPIC206154641.csv: time, flux), stored
                 as arrays for tests/pins.py (run by test_power_host.py and test_gpu_power.py).

The reference's pure-Python inner loops are slow (~0.3 s per period at N=720), so
the cases are small; full-size parity runs against the C oracle, which is itself
pinned by these files (tests/test_oracle_golden.py).

Usage: python tools/gen_golden.py            (takes a few minutes)
"""
import json
import os
import sys
import time
import warnings

import numpy

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, REPO)
import ref_shim  # noqa: E402

ref = ref_shim.activate()
from transitleastsquares import transitleastsquares as ref_tls  # noqa: E402
from transitleastsquares import period_grid as ref_period_grid  # noqa: E402
from transitleastsquares import duration_grid as ref_duration_grid  # noqa: E402
from transitleastsquares.core import search_period as ref_search_period  # noqa: E402
from transitleastsquares.transit import get_cache as ref_get_cache  # noqa: E402
from transitleastsquares.validate import validate_args as ref_validate_args  # noqa: E402
import transitleastsquares.tls_constants as ref_constants  # noqa: E402

from tls_amd import transit_model  # noqa: E402  (data generator only)

OUT = os.path.join(REPO, "tests", "golden")
os.makedirs(OUT, exist_ok=True)


def flatten_table(overview, rows):
    length = numpy.array([len(r) for r in rows], dtype=numpy.int64)
    offset = numpy.concatenate([[0], numpy.cumsum(length)[:-1]]).astype(numpy.int64)
    return dict(tmpl_values=numpy.concatenate([numpy.asarray(r, dtype=float) for r in rows]),
                tmpl_offset=offset, tmpl_length=length,
                tmpl_width=numpy.asarray(overview["width_in_samples"], dtype=numpy.int64),
                tmpl_overshoot=numpy.asarray(overview["overshoot"], dtype=float),
                tmpl_duration=numpy.asarray(overview["duration"], dtype=float))


def reference_plan(t, y, dy, **kwargs):
    """Run the reference's own validation, grids and template table."""
    model = ref_tls(t, y, dy, verbose=False)
    kwargs = dict(kwargs, verbose=False)
    ref_validate_args(model, kwargs)
    periods = ref_period_grid(R_star=model.R_star, M_star=model.M_star,
                              time_span=numpy.max(model.t) - numpy.min(model.t),
                              period_min=model.period_min, period_max=model.period_max,
                              oversampling_factor=model.oversampling_factor,
                              n_transits_min=model.n_transits_min)
    durations = ref_duration_grid(periods, shortest=1 / len(model.t),
                                  log_step=model.duration_grid_step)
    maxwidth = int(numpy.max(durations) * numpy.size(model.y))
    if maxwidth % 2 != 0:
        maxwidth += 1
    overview, rows = ref_get_cache(durations=durations, maxwidth_in_samples=maxwidth,
                                   per=model.per, rp=model.rp, a=model.a, inc=model.inc,
                                   ecc=model.ecc, w=model.w, u=model.u,
                                   limb_dark=model.limb_dark, verbose=False)
    return model, numpy.sort(periods), durations, overview, rows


def search_case(name, t, y, dy, stride, **kwargs):
    t0 = time.time()
    model, periods, durations, overview, rows = reference_plan(t, y, dy, **kwargs)
    sel = periods[::stride]
    out = [ref_search_period(period=p, t=model.t, y=model.y, dy=model.dy,
                             transit_depth_min=model.transit_depth_min,
                             R_star_min=model.R_star_min, R_star_max=model.R_star_max,
                             M_star_min=model.M_star_min, M_star_max=model.M_star_max,
                             lc_arr=rows, lc_cache_overview=overview,
                             T0_fit_margin=model.T0_fit_margin) for p in sel]
    data = dict(t=model.t, y=model.y, dy=model.dy, periods=sel,
                params=numpy.array([model.transit_depth_min, model.R_star_min, model.R_star_max,
                                    model.M_star_min, model.M_star_max, model.T0_fit_margin]),
                chi2=numpy.array([o[1] for o in out], dtype=float),
                row=numpy.array([o[2] for o in out], dtype=numpy.int64),
                depth=numpy.array([o[3] for o in out], dtype=float),
                kwargs_json=numpy.array(json.dumps(kwargs)))
    data.update(flatten_table(overview, rows))
    numpy.savez_compressed(os.path.join(OUT, "search_%s.npz" % name), **data)
    print("search_%s: N=%d, %d of %d periods, %.0fs, chi2 range %.6f..%.6f"
          % (name, len(model.t), len(sel), len(periods), time.time() - t0,
             data["chi2"].min(), data["chi2"].max()), flush=True)


SCALAR_KEYS = ("SDE", "SDE_raw", "chi2_min", "chi2red_min", "period", "period_uncertainty", "T0",
               "duration", "depth", "rp_rs", "snr", "odd_even_mismatch", "transit_count",
               "distinct_transit_count", "empty_transit_count", "FAP", "in_transit_count",
               "after_transit_count", "before_transit_count")
ARRAY_KEYS = ("depth_mean", "depth_mean_even", "depth_mean_odd", "transit_depths",
              "transit_depths_uncertainties", "snr_per_transit", "snr_pink_per_transit",
              "transit_times", "per_transit_count", "periods", "power", "power_raw", "SR", "chi2",
              "chi2red", "model_lightcurve_time", "model_lightcurve_model", "model_folded_phase",
              "folded_y", "folded_dy", "folded_phase", "model_folded_model")


def power_case(name, t, y, dy, **kwargs):
    t0 = time.time()
    numpy.random.seed(1234)  # the reference consumes the global RNG (main.py:129-130)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        res = ref_tls(t, y, dy, verbose=False).power(use_threads=1, show_progress_bar=False,
                                                     **kwargs)
    rng_after = numpy.random.random()  # RNG side effect marker
    data = dict(in_t=numpy.asarray(t, dtype=float), in_y=numpy.asarray(y, dtype=float),
                in_dy=numpy.zeros(0) if dy is None else numpy.asarray(dy, dtype=float),
                kwargs_json=numpy.array(json.dumps(kwargs)), rng_after=rng_after)
    for k in SCALAR_KEYS:
        data["res_" + k] = numpy.float64(res[k])
    for k in ARRAY_KEYS:
        data["res_" + k] = numpy.asarray(res[k], dtype=float)
    numpy.savez_compressed(os.path.join(OUT, "power_%s.npz" % name), **data)
    print("power_%s: %.0fs period=%.8f SDE=%.6f T0=%.8f chi2_min=%.8f"
          % (name, time.time() - t0, res.period, res.SDE, res.T0, res.chi2_min), flush=True)


def injected(n_days, per_day, sigma, seed, per, rp, a, t0_offset=1.0, start=3.14):
    numpy.random.seed(seed)
    n = int(n_days * per_day)
    t = numpy.linspace(start, start + n_days, n)
    sig = transit_model.light_curve(t, start + t0_offset, per, rp, a, 90, 0, 90, [0.4, 0.4],
                                    "quadratic")
    return t, sig + numpy.random.normal(0, sigma, n)


def main():
    # ---- grids and template tables
    grids = {}
    sets = [dict(R_star=1, M_star=1, time_span=0.1), dict(R_star=1, M_star=1, time_span=20),
            dict(R_star=5, M_star=1, time_span=20, period_min=0, period_max=999,
                 oversampling_factor=3),
            dict(R_star=0.5, M_star=0.4, time_span=27, period_min=0.3, oversampling_factor=5,
                 n_transits_min=3),
            dict(R_star=1, M_star=1, time_span=90)]
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for i, kw in enumerate(sets):
            p = ref_period_grid(**kw)
            d = ref_duration_grid(p, shortest=2, log_step=1.05 if i % 2 else 1.1)
            grids["pg%d_kwargs" % i] = numpy.array(json.dumps(kw))
            grids["pg%d_periods" % i] = p
            grids["pg%d_log_step" % i] = 1.05 if i % 2 else 1.1
            grids["pg%d_durations" % i] = numpy.array(d)
    for tag, preset in (("default", {}), ("grazing", dict(transit_template="grazing")),
                        ("box", dict(transit_template="box"))):
        t, y = injected(30, 24, 2e-4, 0, 4.321, 0.05, 12)
        model, periods, durations, overview, rows = reference_plan(t, y, None, **preset)
        for k, v in flatten_table(overview, rows).items():
            grids["tmpl_%s_%s" % (tag, k)] = v
    numpy.savez_compressed(os.path.join(OUT, "grids.npz"), **grids)
    print("grids.npz written", flush=True)

    # ---- search_period goldens (pure-Python reference inner loops)
    t, y = injected(30, 24, 2e-4, 0, 4.321, 0.05, 12)
    search_case("small", t, y, None, 3, period_min=3.5, period_max=5.5, oversampling_factor=2)
    # per-point uncertainties (general-weight path), two noise levels
    numpy.random.seed(5)
    dy = numpy.full(len(y), 2e-4)
    dy[400:] = 6e-4
    y2 = y.copy()
    y2[400:] += numpy.random.normal(0, 5e-4, len(y) - 400)
    search_case("weights", t, y2, dy, 5, period_min=3.5, period_max=5.5, oversampling_factor=2)
    # dense cadence: long windows, T0 stride > 1 (T0_fit_margin 0.1 -> xth up to 17)
    t3, y3 = injected(12, 96, 3e-4, 2, 3.0, 0.06, 9)
    search_case("stride", t3, y3, None, 9, period_min=2.5, period_max=4.0, T0_fit_margin=0.1)
    # T0_fit_margin = 0: every cadence
    search_case("margin0", t, y, None, 11, period_min=3.5, period_max=5.5, oversampling_factor=2,
                T0_fit_margin=0)
    # nothing passes the depth threshold: chi2 == N branch
    search_case("nofit", t, y, None, 12, period_min=3.5, period_max=5.5, oversampling_factor=2,
                transit_depth_min=0.05)
    # gap + unsorted duplicate time stamps (stable sort matters)
    t4, y4 = injected(25, 24, 2e-4, 3, 2.7, 0.05, 10)
    keep = numpy.ones(len(t4), dtype=bool)
    keep[150:260] = False
    t4, y4 = t4[keep], y4[keep]
    t4[40] = t4[39]
    t4[300] = t4[299]
    search_case("gap_ties", t4, y4, None, 6, period_min=2.0, period_max=3.5, oversampling_factor=2)

    # ---- full power() goldens
    power_case("small", t, y, None, period_min=3.5, period_max=5.5, oversampling_factor=2)
    power_case("weights", t, y2, dy, period_min=3.5, period_max=5.5, oversampling_factor=2)
    power_case("nofit", t, y, None, period_min=3.5, period_max=5.5, oversampling_factor=2,
               transit_depth_min=0.05)

    # ---- the reference's own data fixtures, as arrays
    for epic in ("EPIC201367065", "EPIC206154641"):
        d = numpy.genfromtxt(os.path.join(ref_shim.REFERENCE_ROOT, "transitleastsquares", "tests",
                                          epic + ".csv"), delimiter=",", dtype="f8, f8",
                             names=["t", "y"])
        numpy.savez_compressed(os.path.join(OUT, "k2_%s.npz" % epic), t=d["t"], y=d["y"])
    print("done; reference version", ref_constants.TLS_VERSION)


if __name__ == "__main__":
    main()
