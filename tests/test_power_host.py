"""Host logic of power() (spectra, T0 fit, statistics, results object) with the period
search INJECTED from the CPU oracle -- no GPU needed.  The product never does this:
tls_amd.search.search_periods only knows the HIP library; the injection below is a
pytest monkeypatch.  The same checks run on the real HIP path in test_gpu_power.py."""
import numpy
import pytest

import tls_amd
import pins


@pytest.fixture()
def make_model(monkeypatch, oracle_lib):
    def oracle_search_periods(t, y, dy, periods, table, transit_depth_min, R_star_min, R_star_max,
                              M_star_min, M_star_max, T0_fit_margin, **_unused):
        chi2, row, depth, _ = oracle_lib.search(t, y, dy, periods, table, transit_depth_min,
                                                R_star_min, R_star_max, M_star_min, M_star_max,
                                                T0_fit_margin)
        return chi2, row, depth

    def host_t0_fit_residuals(t, y, period, signal, T0_array, roll, **_unused):
        return oracle_lib.t0_residuals(t, y, period, signal, T0_array, roll)   # oracle: stats.py:178-195

    def oracle_spectra(chi2, oversampling_factor, **_unused):
        return oracle_lib.spectra(chi2, int(oversampling_factor * 30))   # oracle: stats.py:105-132

    monkeypatch.setattr(tls_amd.search, "spectra", oracle_spectra)
    monkeypatch.setattr(tls_amd.search, "search_periods", oracle_search_periods)
    monkeypatch.setattr(tls_amd.search, "t0_fit_residuals", host_t0_fit_residuals)
    return lambda t, y, dy: tls_amd.transitleastsquares(t, y, dy, verbose=False)


@pytest.mark.parametrize("name", ["small", "weights", "nofit"])
def test_power_matches_reference_golden(make_model, name):
    pins.check_power_golden(make_model, name)


def test_reference_pin_synthetic(make_model):
    pins.check_synthetic(make_model)


def test_reference_pin_transit_depth_min(make_model):
    pins.check_transit_depth_min(make_model)


def test_reference_pin_uncertainties(make_model):
    pins.check_uncertainties(make_model)


def test_reference_pin_stats_gap(make_model):
    pins.check_stats_gap(make_model)


def test_reference_pin_shapes(make_model):
    pins.check_shapes(make_model)


def test_product_search_has_no_cpu_path():
    """Without a GPU the product path must raise, not fall back."""
    from tls_amd import _lib
    if _lib.load().tls_device_count() > 0:
        pytest.skip("a GPU is present")
    t = numpy.linspace(0, 30, 500)
    y = numpy.ones(500)
    y[::40] = 0.999
    with pytest.raises(RuntimeError):
        tls_amd.transitleastsquares(t, y, verbose=False).power(verbose=False,
                                                              show_progress_bar=False)
    # ... and the final T0 fit has no host evaluation either
    from tls_amd.stats import final_T0_fit
    with pytest.raises(RuntimeError):
        final_T0_fit(numpy.full(5, 0.9), 0.999, t, y, None, 3.0, 0.01, False, False, None)
