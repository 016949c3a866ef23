"""The drop-in API end to end on the GPU: tls_amd.transitleastsquares(...).power(...)
against golden results of the unmodified reference and the reference's own
known-answer tests (same checks as test_power_host.py, real HIP search)."""
import numpy
import pytest

import tls_amd
import pins

pytestmark = pytest.mark.gpu


def make_model(t, y, dy):
    return tls_amd.transitleastsquares(t, y, dy, verbose=False)


@pytest.mark.parametrize("name", ["small", "weights", "nofit"])
def test_power_matches_reference_golden(name):
    pins.check_power_golden(make_model, name)


def test_reference_pin_synthetic():
    pins.check_synthetic(make_model)


def test_reference_pin_transit_depth_min():
    pins.check_transit_depth_min(make_model)


def test_reference_pin_uncertainties():
    pins.check_uncertainties(make_model)


def test_reference_pin_stats_gap():
    pins.check_stats_gap(make_model)


def test_reference_pin_multi_planet():
    pins.check_multi_planet(make_model)


def test_reference_pin_shapes():
    pins.check_shapes(make_model)


def test_k2_90d_anchor_values():
    """SURVEY.md Appendix D, config 2 (shimmed reference + C inner loops)."""
    from tls_amd import synthetic
    t, f, kw = synthetic.config("k2_90d")
    r = make_model(t, f, None).power(verbose=False, show_progress_bar=False, **kw)
    assert len(r.periods) == 9679
    assert int(numpy.argmin(r.chi2)) == 7738 == int(numpy.argmax(r.power))
    numpy.testing.assert_allclose(r.chi2_min, 4101.6439079674, rtol=1e-10)
    numpy.testing.assert_allclose(numpy.sum(r.chi2), 41768414.394605, rtol=1e-10)
    numpy.testing.assert_allclose(r.period, 10.12452360, rtol=1e-8)
    numpy.testing.assert_allclose(r.T0, 13.25983524, rtol=1e-8)
    numpy.testing.assert_allclose(r.duration, 0.18549107, rtol=1e-7)
    numpy.testing.assert_allclose(r.depth, 0.99989805, rtol=1e-7)
    numpy.testing.assert_allclose(r.SDE, 25.71647715, rtol=1e-7)
    numpy.testing.assert_allclose(r.SDE_raw, 22.94488813, rtol=1e-7)
    numpy.testing.assert_allclose(r.rp_rs, 0.00908039, rtol=1e-6)
    numpy.testing.assert_allclose(r.snr, 14.324393, rtol=1e-6)
    assert len(r.transit_times) == 8
