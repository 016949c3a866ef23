"""Host-side restatements (grids, template table, helpers, FAP, validation) against
the reference's own unit-test constants and against golden outputs of the
unmodified reference (tests/golden/grids.npz)."""
import json
import os
import warnings

import numpy
import pytest

import tls_amd
from tls_amd import period_grid, duration_grid, resample, cleaned_array, FAP, transit_mask
from tls_amd.template import get_cache, TemplateTable
from tls_amd import constants as C
from conftest import GOLDEN


def test_period_grid_known_answers():
    # constants of the reference's tests/test_period_grid.py:8-50
    periods = period_grid(R_star=1, M_star=1, time_span=0.1)
    numpy.testing.assert_almost_equal(max(periods), 2.4999999999999987)
    numpy.testing.assert_almost_equal(min(periods), 0.6002621413799498)
    assert len(periods) == 268
    periods = period_grid(R_star=1, M_star=1, time_span=20)
    numpy.testing.assert_almost_equal(max(periods), 10)
    numpy.testing.assert_almost_equal(min(periods), 0.6015575922909607)
    assert len(periods) == 1716
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        periods = period_grid(R_star=5, M_star=1, time_span=20, period_min=0, period_max=999,
                              oversampling_factor=3)
    numpy.testing.assert_almost_equal(max(periods), 10)
    numpy.testing.assert_almost_equal(min(periods), 0.6015575922909607)
    assert len(periods) == 1716


def test_period_grid_large():
    periods = period_grid(R_star=0.1, M_star=1, time_span=1000, period_min=0, period_max=999,
                          oversampling_factor=3)
    assert len(periods) == 4308558  # tests/test_period_grid.py:49


def test_duration_grid_known_answers():
    # tests/test_duration_grid.py:7-18
    periods = period_grid(R_star=1, M_star=1, time_span=20, period_min=0, period_max=999,
                          oversampling_factor=3)
    durations = duration_grid(periods, log_step=1.05, shortest=2)
    numpy.testing.assert_almost_equal(max(durations), 0.12)
    numpy.testing.assert_almost_equal(min(durations), 0.004562690993268325)
    assert len(durations) == 69


def test_grids_bit_identical_to_reference_outputs():
    g = numpy.load(os.path.join(GOLDEN, "grids.npz"))
    i = 0
    while "pg%d_kwargs" % i in g:
        kw = json.loads(str(g["pg%d_kwargs" % i]))
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            p = period_grid(**kw)
        numpy.testing.assert_array_equal(p, g["pg%d_periods" % i])
        d = duration_grid(p, shortest=2, log_step=float(g["pg%d_log_step" % i]))
        numpy.testing.assert_array_equal(numpy.array(d), g["pg%d_durations" % i])
        i += 1
    assert i >= 4


@pytest.mark.parametrize("tag,preset", [("default", {}), ("grazing", {"transit_template": "grazing"}),
                                        ("box", {"transit_template": "box"})])
def test_template_table_bit_identical_to_reference(tag, preset):
    """get_cache output (transit.py:98-160) incl. the Mandel-Agol restatement."""
    from tls_amd import synthetic
    g = numpy.load(os.path.join(GOLDEN, "grids.npz"))
    numpy.random.seed(0)
    n = 30 * 24
    t = numpy.linspace(3.14, 33.14, n)
    y = numpy.ones(n)
    y[::50] = 0.999  # the table does not depend on the flux values
    inp = synthetic.search_inputs(t, y, **preset)
    tab = inp["table"]
    numpy.testing.assert_array_equal(tab.width, g["tmpl_%s_tmpl_width" % tag])
    numpy.testing.assert_array_equal(tab.length, g["tmpl_%s_tmpl_length" % tag])
    numpy.testing.assert_array_equal(tab.offset, g["tmpl_%s_tmpl_offset" % tag])
    numpy.testing.assert_array_equal(tab.duration, g["tmpl_%s_tmpl_duration" % tag])
    numpy.testing.assert_array_equal(tab.values, g["tmpl_%s_tmpl_values" % tag])
    numpy.testing.assert_array_equal(tab.overshoot, g["tmpl_%s_tmpl_overshoot" % tag])


def test_default_template_shape_facts():
    # 1742 of 10000 supersamples in transit, first at 4129, depth 1.0843e-3
    from tls_amd import transit_model
    t = numpy.linspace(-0.5, 0.5, C.SUPERSAMPLE_SIZE)
    flux = transit_model.light_curve(t, 0, C.DEFAULT_PERIOD, C.DEFAULT_RP, C.DEFAULT_A,
                                     C.DEFAULT_INC, 0, 90, C.DEFAULT_U, "quadratic")
    assert int(numpy.argmax(flux < 1)) == 4129
    assert int(numpy.sum(flux < 1)) == 1742
    numpy.testing.assert_allclose(1 - flux.min(), 1.0843e-3, rtol=2e-4)


def test_transit_model_limits():
    from tls_amd.transit_model import quadratic_ld_flux, ellip_k, ellip_e
    # uniform source, planet fully inside: depth = p^2
    z = numpy.array([0.0, 0.2, 0.5])
    numpy.testing.assert_allclose(1 - quadratic_ld_flux(z, 0.1, 0.0, 0.0), 0.01, rtol=1e-7)
    # out of transit
    assert numpy.all(quadratic_ld_flux(numpy.array([1.2, 5.0, 1e10]), 0.1, 0.4, 0.2) == 1.0)
    # limb darkening: centre deeper than limb, flux continuous across contacts
    f = quadratic_ld_flux(numpy.array([0.0, 0.85, 0.8999, 0.9001, 1.0999, 1.1001]), 0.1, 0.4, 0.3)
    assert f[0] < f[1] < 1
    assert abs(f[2] - f[3]) < 1e-5 and abs(f[4] - f[5]) < 1e-5
    # polynomial elliptic integrals are close to the exact ones (2e-8)
    from scipy import special
    k = numpy.array([0.1, 0.5, 0.9])
    numpy.testing.assert_allclose(ellip_k(k), special.ellipk(k ** 2), atol=5e-8)
    numpy.testing.assert_allclose(ellip_e(k), special.ellipe(k ** 2), atol=5e-8)


def test_resample_known_answers():
    # tests/test_resample.py:9-45
    testtime = numpy.linspace(0, 1, 1000)
    testflux = numpy.linspace(0.99, 1.01, 1000)
    a, b = resample(time=testtime, flux=testflux, factor=100)
    numpy.testing.assert_almost_equal(a, numpy.linspace(0, 1, 10))
    numpy.testing.assert_almost_equal(
        b, (0.99, 0.99222222, 0.99444444, 0.99666667, 0.99888889, 1.00111111, 1.00333333,
            1.00555556, 1.00777778, 1.01))
    assert len(a) == 10 and len(b) == 10


def test_cleaned_array_known_answers():
    # tests/test_cleaned_array.py:8-22
    dirty = numpy.ones(10, dtype=object)
    time_array = numpy.linspace(1, 10, 10)
    dy_array = numpy.ones(10, dtype=object)
    dirty[1] = None
    dirty[2] = numpy.inf
    dirty[3] = -numpy.inf
    dirty[4] = numpy.nan
    dirty[5] = -99
    time_array[8] = numpy.nan
    dy_array[9] = numpy.inf
    t, y, dy = cleaned_array(time_array, dirty, dy_array)
    numpy.testing.assert_equal(t, [1, 7, 8])
    numpy.testing.assert_equal(y, [1, 1, 1])
    numpy.testing.assert_equal(dy, [1, 1, 1])
    t2, y2 = cleaned_array(time_array, dirty)
    numpy.testing.assert_equal(t2, [1, 7, 8, 10])


def test_fap_known_answers():
    # tests/test_FAP.py:7-9
    numpy.testing.assert_equal(FAP(SDE=2), numpy.nan)
    numpy.testing.assert_equal(FAP(SDE=7), 0.009443778)
    numpy.testing.assert_equal(FAP(SDE=99), 8.0032e-05)


def test_validation_errors():
    # tests/test_validation.py:7-18 plus the ValueErrors of validate.py:21-44,122-175
    t = y = numpy.linspace(0.5, 1.5, 2000)
    for bad in (0, "1"):
        with pytest.raises(ValueError):
            tls_amd.transitleastsquares(t, y).power(use_threads=bad)
    with pytest.raises(ValueError):
        tls_amd.transitleastsquares(t, y).power(period_min=5, period_max=2)
    with pytest.raises(ValueError):
        tls_amd.transitleastsquares(t, y).power(transit_template="nope")
    with pytest.raises(ValueError):
        tls_amd.transitleastsquares(t, y).power(R_star=-1)
    with pytest.raises(ValueError):
        tls_amd.transitleastsquares(t, y).power(n_transits_min=1.5)
    with pytest.raises(ValueError):
        tls_amd.transitleastsquares(numpy.ones(10), numpy.ones(10))  # zero time span
    with pytest.raises(ValueError):
        tls_amd.transitleastsquares(numpy.array([1.0, 2.0]), numpy.array([1.0, 1.0]))
    with pytest.warns(UserWarning):
        tls_amd.transitleastsquares(numpy.linspace(1, 2, 50), numpy.full(50, 3.0))


def test_inputs_normalisation():
    t = numpy.linspace(1, 10, 100)
    y = 1 + 1e-3 * numpy.sin(t)
    dy = numpy.full(100, 0.002)
    m = tls_amd.transitleastsquares(t, y, dy)
    numpy.testing.assert_allclose(m.dy, 1.0)  # dy / mean(dy), validate.py:18
    m = tls_amd.transitleastsquares(t, y)
    numpy.testing.assert_allclose(m.dy, numpy.std(y))  # validate.py:39-40


def test_results_object_contract():
    from tls_amd.results import transitleastsquaresresults, RESULT_KEYS
    assert len(RESULT_KEYS) == 41
    r = transitleastsquaresresults(*range(41))
    assert list(r.keys())[:5] == ["SDE", "SDE_raw", "chi2_min", "chi2red_min", "period"]
    assert r.SDE == 0 and r["model_folded_model"] == 40 and r.chi2 == 32
    with pytest.raises(AttributeError):
        r.nope


def test_transit_mask_and_public_surface():
    t = numpy.linspace(0, 10, 101)
    m = transit_mask(t, 5.0, 1.0, 2.5)
    assert m[25] and m[75] and not m[0]
    for name in ("transitleastsquares", "cleaned_array", "resample", "transit_mask",
                 "duration_grid", "period_grid", "FAP", "fold"):
        assert hasattr(tls_amd, name)


def test_pink_noise_equals_reference_loop():
    """Vectorised pink_noise against the reference's loop (stats.py:72-77), bit for bit."""
    from tls_amd.stats import pink_noise
    rng = numpy.random.RandomState(0)
    for n, w in ((4000, 7), (3000, 13), (500, 1), (200, 200), (1000, 64)):
        d = 1 + rng.normal(0, 1e-4, n)
        total = 0
        for i in range(n - w + 1):
            total += numpy.std(d[i: i + w]) / w ** 0.5
        assert pink_noise(d, w) == total / (n - w + 1)


def test_running_median_fast_path_is_identical():
    from tls_amd.helpers import running_median
    rng = numpy.random.RandomState(1)
    for n, k in ((500, 91), (9679, 91), (200, 31), (120, 30), (64, 7)):
        data = rng.normal(0, 1, n)
        idx = numpy.arange(k) + numpy.arange(n - k + 1)[:, None]        # reference helpers.py:95
        med = numpy.median(data[idx], axis=1)
        missing = n - len(med)
        front = int(missing * 0.5)
        want = numpy.concatenate([numpy.full(front, med[0]), med, numpy.full(missing - front, med[-1])])
        numpy.testing.assert_array_equal(running_median(data, k), want)


def test_transit_model_other_laws_and_eccentric_orbits():
    """The laws without a closed form (batman: nonlinear, squareroot, logarithmic, exponential, power2)
    are integrated numerically; with coefficients that reduce them to the quadratic / linear law they
    must reproduce the closed form (to the 2e-8 of its Hastings polynomials).  Eccentric orbits: the
    transit stays centred on t0, e -> 0 is the circular orbit, and the template builder accepts them
    (the reference hands ecc, w, u, limb_dark to batman under transit_template='default',
    transit.py:14-25)."""
    from tls_amd import transit_model as tm
    t = numpy.linspace(-0.3, 0.3, 2001)
    args = (0.0, 12.9, 0.03, 23.1, 89.21)
    u1, u2 = 0.4804, 0.1867
    quad = tm.light_curve(t, *args, 0, 90, [u1, u2], "quadratic")
    nonl = tm.light_curve(t, *args, 0, 90, [0.0, u1 + 2 * u2, 0.0, -u2], "nonlinear")
    numpy.testing.assert_allclose(nonl, quad, rtol=0, atol=3e-8)
    lin = tm.light_curve(t, 0.0, 12.9, 0.1, 23.1, 89.5, 0, 90, [0.5], "linear")
    for law, u in (("squareroot", [0.5, 0.0]), ("power2", [0.5, 1.0]), ("logarithmic", [0.5, 0.0])):
        numpy.testing.assert_allclose(tm.light_curve(t, 0.0, 12.9, 0.1, 23.1, 89.5, 0, 90, u, law), lin, rtol=0, atol=3e-8)
    grazing = tm.light_curve(t, 0.0, 12.9, 0.2, 10.0, 85.0, 0, 90, [0.4, 0.3], "quadratic")
    numpy.testing.assert_allclose(tm.light_curve(t, 0.0, 12.9, 0.2, 10.0, 85.0, 0, 90, [0.0, 1.0, 0.0, -0.3], "nonlinear"),
                                  grazing, rtol=0, atol=3e-8)
    with pytest.raises(ValueError):
        tm.light_curve(t, *args, 0, 90, [0.1], "nonlinear")
    # eccentric orbit: deepest point at the conjunction t0, shorter transit near periastron than the circular one
    ecc = tm.light_curve(t, 0.0, 12.9, 0.1, 23.1, 89.5, 0.3, 90, [0.5], "linear")
    assert abs(t[numpy.argmin(ecc)]) < 2 * (t[1] - t[0])
    assert 0 < (ecc < 1).sum() < (lin < 1).sum()
    numpy.testing.assert_allclose(tm.light_curve(t, 0.0, 12.9, 0.1, 23.1, 89.5, 1e-7, 90, [0.5], "linear"), lin, atol=1e-7)
    with pytest.raises(ValueError):
        tm.projected_separation(t, 0.0, 12.9, 23.1, 89.5, 1.2, 90)
    # through the template builder, as power(limb_dark=..., u=..., ecc=..., w=...) would
    from tls_amd import synthetic
    tt, f = synthetic.light_curve(20.0, 24, 2e-4, per=4.0, rp=0.05, a=12)
    inp = synthetic.search_inputs(tt, f, period_min=3.0, period_max=5.0, limb_dark="nonlinear", u=[0.1, 0.5, 0.1, -0.1],
                                  ecc=0.2, w=60)
    assert inp["table"].n_rows > 3 and numpy.all(numpy.isfinite(inp["table"].values))


def test_pink_noise_dispatch_keeps_small_inputs_on_the_host():
    """search.pink_noise (what power() hands to snr_stats): inputs below PINK_NOISE_ON_DEVICE window elements, and the inputs
    the reference itself fails on, never reach the device -- the numpy form answers (no GPU in this test) -- and
    snr_stats takes the function it is handed."""
    from tls_amd import search, stats
    rng = numpy.random.RandomState(2)
    data = 1 + rng.normal(0, 1e-3, 900)
    assert search.pink_noise(data, 9) == stats.pink_noise(data, 9)
    with pytest.raises(ValueError):
        search.pink_noise(data[:5], 9)
    nan = data.copy(); nan[3] = numpy.nan
    assert numpy.isnan(search.pink_noise(nan, 9))
    calls = []

    def fn(d, w):
        calls.append((len(d), w))
        return 1e-4
    t = numpy.linspace(0, 30, 1440)
    y = 1 + rng.normal(0, 1e-3, len(t))
    transit_times = [5.0, 15.0, 25.0]
    snr, snr_pink = stats.snr_stats(t, y, 10.0, 0.01, 5.0, transit_times, 0.3, numpy.array([14, 15, 14]), pink_noise_fn=fn)
    assert calls and calls[0][1] == 14 and len(snr) == len(snr_pink) == 3
