"""Shared checks: full power() results against (a) golden outputs of the unmodified
reference and (b) the known answers hard-coded in the reference's own tests.

Each check takes `make_model(t, y, dy)`, so the same assertions run with the HIP
search (GPU tests) and with the search injected from the CPU oracle (host-logic
tests, no GPU)."""
import os
import warnings

import numpy

from tls_amd import transit_model, transit_mask, cleaned_array
from conftest import GOLDEN, load_power_golden

SCALARS = ("SDE", "SDE_raw", "chi2_min", "chi2red_min", "period", "period_uncertainty", "T0",
           "duration", "depth", "rp_rs", "snr", "odd_even_mismatch", "transit_count",
           "distinct_transit_count", "empty_transit_count", "FAP", "in_transit_count",
           "after_transit_count", "before_transit_count")
ARRAYS = ("depth_mean", "depth_mean_even", "depth_mean_odd", "transit_depths",
          "transit_depths_uncertainties", "snr_per_transit", "snr_pink_per_transit",
          "transit_times", "per_transit_count", "periods", "power", "power_raw", "SR", "chi2",
          "chi2red", "model_lightcurve_time", "model_lightcurve_model", "model_folded_phase",
          "folded_y", "folded_dy", "folded_phase", "model_folded_model")


def check_power_golden(make_model, name):
    """All 41 result fields against power_<name>.npz (reference outputs)."""
    g, t, y, dy, kwargs = load_power_golden(name)
    numpy.random.seed(1234)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        res = make_model(t, y, dy).power(use_threads=1, show_progress_bar=False, verbose=False,
                                         **kwargs)
    # same consumption of the global RNG as the reference (main.py:129-130)
    assert numpy.random.random() == float(g["rng_after"])
    assert len(res) == 41
    for k in SCALARS:
        numpy.testing.assert_allclose(float(res[k]), float(g["res_" + k]), rtol=1e-9, atol=1e-12,
                                      err_msg=k)
    for k in ARRAYS:
        numpy.testing.assert_allclose(numpy.asarray(res[k], dtype=float), g["res_" + k],
                                      rtol=1e-9, atol=1e-11, err_msg=k)
    # argmin/argmax indices are exact
    assert int(numpy.argmin(res.chi2)) == int(numpy.argmin(g["res_chi2"]))
    assert int(numpy.argmax(res.power)) == int(numpy.argmax(g["res_power"]))
    return res


def _three_year_curve(gap=False, excess_noise=False):
    """Data generator of the reference's tests/test_synthetic.py:9-38 (and variants)."""
    numpy.random.seed(seed=0)
    start, days, samples_per_day = 48, 365.25 * 3, 12
    samples = int(days * samples_per_day)
    t = numpy.linspace(start, start + days, samples)
    flux = transit_model.light_curve(t, start + 20, 365.25, 6371 / 696342, 217, 90, 0, 90, [0.5],
                                     "linear")
    stdev = 10 ** -6 * 5
    y = flux + numpy.random.normal(0, stdev, int(samples))
    dy = None
    if excess_noise:  # tests/test_uncertainties.py:38-46
        y[10000:] = y[10000:] + numpy.random.normal(0, 10 * stdev, 3149)
        dy = numpy.full(len(y), stdev)
        dy[10000:] = 10 * stdev
    else:
        y[1] = numpy.nan
    if gap:  # tests/test_stats_gap.py:41-42
        y[200:500] = numpy.nan
        t[200:500] = numpy.nan
    return t, y, dy


def check_synthetic(make_model):
    """tests/test_synthetic.py:40-63"""
    t, y, _ = _three_year_curve()
    results = make_model(t, y, None).power(
        period_min=360, period_max=370, transit_depth_min=10 * 10 ** -6, oversampling_factor=5,
        duration_grid_step=1.02, verbose=False, use_threads=1, show_progress_bar=False)
    numpy.testing.assert_almost_equal(results.chi2_min, 8831.654060613922, decimal=5)
    numpy.testing.assert_almost_equal(results.chi2red_min, 0.6719152511118321, decimal=5)
    numpy.testing.assert_almost_equal(results.period_uncertainty, 0.216212529678387, decimal=5)
    numpy.testing.assert_equal(results.per_transit_count[0], 7)
    numpy.testing.assert_equal(len(results.transit_times), 3)
    numpy.testing.assert_almost_equal(results.period, 365.2582192473641, decimal=5)
    numpy.testing.assert_almost_equal(results.transit_times[0], 68.00349264912924, decimal=5)


def check_transit_depth_min(make_model):
    """tests/test_transit_depth_min.py:41-71 (nothing is fit)"""
    numpy.random.seed(0)  # the reference test leaves the RNG unseeded; the outcome does not depend on it
    t, y, _ = _three_year_curve()
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        results = make_model(t, y, None).power(
            transit_depth_min=1000 * 10 ** -6, period_min=360, period_max=370,
            oversampling_factor=5, duration_grid_step=1.02, T0_fit_margin=0.1, verbose=False,
            show_progress_bar=False)
    for key in ("transit_times", "period", "duration", "snr", "snr_pink_per_transit",
                "odd_even_mismatch", "in_transit_count", "after_transit_count",
                "before_transit_count"):
        numpy.testing.assert_equal(results[key], numpy.nan)
    numpy.testing.assert_equal(results.depth, 1)
    numpy.testing.assert_equal(results.SDE, 0)
    numpy.testing.assert_equal(results.SDE_raw, 0)
    numpy.testing.assert_almost_equal(results.chi2_min, 13148.0)
    numpy.testing.assert_almost_equal(results.chi2red_min, 1.0003043213633598)
    numpy.testing.assert_equal(len(results.periods), 278)
    numpy.testing.assert_almost_equal(max(results.periods), 369.9831654894093)
    numpy.testing.assert_almost_equal(min(results.periods), 360.0118189140635)
    numpy.testing.assert_almost_equal(max(results.power), 0)
    numpy.testing.assert_almost_equal(min(results.power), 0)
    numpy.testing.assert_almost_equal(max(results.chi2), 13148.0)
    numpy.testing.assert_almost_equal(max(results.chi2red), 1.0003043213633598)
    assert numpy.all(results.chi2 == 13148.0)  # exactly N where nothing was fit (core.py:46)


def check_uncertainties(make_model):
    """tests/test_uncertainties.py:46-57"""
    t, y, dy = _three_year_curve(excess_noise=True)
    results = make_model(t, y, dy).power(
        period_min=360, period_max=370, oversampling_factor=3, duration_grid_step=1.05,
        T0_fit_margin=0.2, verbose=False, show_progress_bar=False)
    numpy.testing.assert_almost_equal(results.SDE, 5.292594615900944, decimal=5)


def check_stats_gap(make_model):
    """tests/test_stats_gap.py:41-132"""
    t, y, _ = _three_year_curve(gap=True)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        results = make_model(t, y, None).power(
            period_min=360, period_max=370, transit_depth_min=10 * 10 ** -6, oversampling_factor=2,
            duration_grid_step=1.1, T0_fit_margin=1.2, verbose=False, show_progress_bar=False)
    aae = numpy.testing.assert_almost_equal
    aae(results.period_uncertainty, 0.3153203546531813, decimal=5)
    numpy.testing.assert_equal(results.per_transit_count, [0, 5, 5])
    numpy.testing.assert_equal(len(results.transit_times), 3)
    aae(results.period, 365.22218620040417, decimal=5)
    aae(results.transit_times, [68.08637, 433.30855, 798.53074], decimal=5)
    aae(results.depth, 0.9998972750356973, decimal=5)
    aae(results.duration, 0.41845319797978703, decimal=5)
    aae(results.SDE, 4.243572802600693, decimal=3)
    aae(results.odd_even_mismatch, 0.15059221218811772, decimal=3)
    aae(results.rp_rs, 0.009114758081257387, decimal=3)
    aae(numpy.sum(results.model_lightcurve_time), 38275494.19583159, decimal=3)
    aae(numpy.sum(results.model_lightcurve_model), 64233.9941755991, decimal=3)
    aae(max(results.model_folded_phase), 1.0000380285975052, decimal=3)
    aae(min(results.model_folded_phase), 3.8028597505324e-05, decimal=3)
    aae(numpy.mean(results.model_folded_phase), 0.5000380285975052, decimal=3)
    aae(results.depth_mean_even, (0.999915, 6.785539e-06), decimal=3)
    aae(results.depth_mean_odd, (0.999920, 1.209993e-05), decimal=3)
    aae(results.depth_mean, (0.999917, 6.086923e-06), decimal=3)
    aae(results.transit_depths, [numpy.nan, 0.99991, 0.9999], decimal=3)
    aae(results.transit_depths_uncertainties, [numpy.nan, 2.92371e-06, 4.48803e-06], decimal=3)
    aae(results.transit_count, 3, decimal=3)
    aae(results.distinct_transit_count, 2, decimal=3)
    aae(results.empty_transit_count, 1, decimal=3)
    aae(results.snr_per_transit, [0., 37.052, 36.558], decimal=3)
    aae(results.snr, 52.050323372452034, decimal=3)
    aae(results.snr_pink_per_transit, [0., 45.477, 44.871], decimal=3)


def _k2(epic):
    from scipy.signal import medfilt
    d = numpy.load(os.path.join(GOLDEN, "k2_%s.npz" % epic))
    t, y = d["t"], d["y"]
    return t, y / medfilt(y, 25)


def check_multi_planet(make_model):
    """tests/test_multi_planet.py:15-49 (K2-3, two passes)"""
    t, y_filt = _k2("EPIC201367065")
    results = make_model(t, y_filt, None).power(verbose=False, show_progress_bar=False)
    aae = numpy.testing.assert_almost_equal
    aae(max(results.power), 45.49085809486116, decimal=3)
    aae(max(results.power_raw), 42.93056655774114, decimal=3)
    aae(min(results.power), -0.6175100139942546, decimal=3)
    aae(min(results.power_raw), -0.3043720539933344, decimal=3)
    intransit = transit_mask(t, results.period, 2 * results.duration, results.T0)
    t2, y2 = cleaned_array(t[~intransit], y_filt[~intransit])
    second = make_model(t2, y2, None).power(verbose=False, show_progress_bar=False)
    aae(second.duration, 0.15061016994013998, decimal=3)
    aae(second.SDE, 34.9911304598618, decimal=3)
    aae(second.rp_rs, 0.025852178872027086, decimal=3)


def check_shapes(make_model):
    """tests/test_shapes.py:17-36 (box and grazing templates)"""
    t, y_filt = _k2("EPIC206154641")
    aae = numpy.testing.assert_almost_equal
    box = make_model(t, y_filt, None).power(transit_template="box", verbose=False,
                                            show_progress_bar=False)
    aae(box.duration, 0.06111785726416931, decimal=5)
    aae(box.rp_rs, 0.08836981203437415, decimal=5)
    grazing = make_model(t, y_filt, None).power(transit_template="grazing", verbose=False,
                                                show_progress_bar=False)
    aae(grazing.duration, 0.08948265482047034, decimal=5)
    aae(min(grazing.chi2red), 0.06759475703796078, decimal=5)
