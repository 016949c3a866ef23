"""Drop-in `transitleastsquares(t, y, dy).power(**kwargs)`.

Same constructor, same keyword set and defaults, same 41-key results object as
the reference (main.py:44-455).  The one thing that differs is WHERE the
period x duration x T0 grid search runs: the reference maps `search_period`
over a multiprocessing pool (main.py:140-185); here the whole period grid goes
to the MI355X in one batched call through the C ABI (tls_amd.search ->
libtls_amd.so, include/tls_amd.h).  There is no CPU fallback: without the HIP
library or a GPU, power() raises.
"""
import warnings

import numpy

from . import constants as C
from . import search as _search
from .grid import duration_grid, period_grid
from .helpers import fold, transit_mask
from .results import transitleastsquaresresults
from .stats import (FAP, all_transit_times, calculate_fill_factor, calculate_stretch,
                    calculate_transit_duration_in_days, count_stats, final_T0_fit,
                    intransit_stats, model_lightcurve, period_uncertainty, rp_rs_from_depth,
                    snr_stats, _intransit_fluxes)
from .template import TemplateTable, fractional_transit, get_cache
from .validate import validate_args, validate_inputs

_NAN = numpy.nan
_GRID_CACHE = {}  # (time span, n, kwargs) -> (periods, durations, lc_cache_overview, lc_arr)


class transitleastsquares(object):
    """Compute the transit least squares of limb-darkened transit models"""

    def __init__(self, t, y, dy=None, verbose=True):
        self.t, self.y, self.dy = validate_inputs(t, y, dy)
        self.verbose = verbose

    # ------------------------------------------------------------------ search
    def _build_grids(self):
        """Period grid, duration grid and template table of this search.  They depend only on
        the time span, the number of points and the keyword arguments, so surveys that call
        power() on many light curves with shared time stamps get them from a small cache."""
        key = (len(self.t), float(numpy.min(self.t)), float(numpy.max(self.t)), self.R_star, self.M_star,
               self.period_min, self.period_max, self.oversampling_factor, self.n_transits_min,
               self.duration_grid_step, self.per, self.rp, self.a, self.inc, self.ecc, self.w,
               tuple(self.u), self.limb_dark)
        hit = _GRID_CACHE.get(key)
        if hit is None:
            hit = self._build_grids_uncached()
            # the ascending period grid (results come back ordered like it, main.py:190-196) and the flattened template table
            # ride in the cache entry: a survey calling power() per light curve forms them once, not per call
            hit = hit + (numpy.sort(numpy.asarray(hit[0], dtype=numpy.float64)), TemplateTable(hit[2], hit[3]))
            if len(_GRID_CACHE) >= 8:
                _GRID_CACHE.pop(next(iter(_GRID_CACHE)))
            _GRID_CACHE[key] = hit
        elif self.verbose:
            print("Creating model cache for", str(len(hit[1])), "durations")
        return hit

    def _build_grids_uncached(self):
        periods = period_grid(
            R_star=self.R_star, M_star=self.M_star,
            time_span=numpy.max(self.t) - numpy.min(self.t),
            period_min=self.period_min, period_max=self.period_max,
            oversampling_factor=self.oversampling_factor,
            n_transits_min=self.n_transits_min)
        durations = duration_grid(periods, shortest=1 / len(self.t),
                                  log_step=self.duration_grid_step)
        maxwidth_in_samples = int(numpy.max(durations) * numpy.size(self.y))
        if maxwidth_in_samples % 2 != 0:
            maxwidth_in_samples = maxwidth_in_samples + 1
        lc_cache_overview, lc_arr = get_cache(
            durations=durations, maxwidth_in_samples=maxwidth_in_samples, per=self.per,
            rp=self.rp, a=self.a, inc=self.inc, ecc=self.ecc, w=self.w, u=self.u,
            limb_dark=self.limb_dark, verbose=self.verbose)
        return periods, durations, lc_cache_overview, lc_arr

    def power(self, **kwargs):
        """Compute the periodogram for a set of user-defined parameters"""
        self, kwargs = validate_args(self, kwargs)
        if self.verbose:
            print(C.BACKEND_BANNER)

        periods, durations, lc_cache_overview, lc_arr, test_statistic_periods, table = self._build_grids()
        if self.verbose:
            print("Searching " + str(len(self.y)) + " data points, " + str(len(periods))
                  + " periods from " + str(round(numpy.min(periods), 3)) + " to "
                  + str(round(numpy.max(periods), 3)) + " days")

        # The reference shuffles the search order with the GLOBAL numpy RNG
        # (main.py:129-130); the GPU does not need an order, but callers that
        # seeded the RNG must find it in the same state afterwards.
        if C.PERIODS_SEARCH_ORDER == "shuffled":
            numpy.random.permutation(periods)

        # one batched device call replaces the pool of main.py:140-185; results come
        # back ordered like the ascending period grid (main.py:190-196)
        # (test_statistic_periods, table: from the grid cache entry)
        # devices: the period grid sharded over several GPUs of this process (tls_amd.search.DeviceGroup: one context and
        # one host thread per device, blocks by modelled time, one RCCL all-gather) -- the counterpart of the reference's
        # use_threads pool over periods (main.py:140-163).  Default "auto", like use_threads = cpu_count()
        # (validate.py:81): every visible GPU when the modelled one-GPU time exceeds the overhead of sharding, one GPU
        # otherwise (search.auto_devices); a list names the GPUs; device= / context= keep the search on one.
        params = dict(transit_depth_min=self.transit_depth_min, R_star_min=self.R_star_min, R_star_max=self.R_star_max,
                      M_star_min=self.M_star_min, M_star_max=self.M_star_max, T0_fit_margin=self.T0_fit_margin)
        # One GPU (the usual case): the whole device part of this call -- search, spectra, pick, final T0 fit -- is ONE
        # submission with one wait at its end (search.fused_power -> tls_power_batch with one light curve).  Several GPUs:
        # the search over the group, then spectra and T0 fit on its first device, as before.  (Tests that inject a CPU
        # search replace search_periods: the fused chain is taken only while that is the product's function.)
        fused = None
        if getattr(_search.search_periods, "_tls_amd_product", False):
            kind, what = _search.resolve_devices(kwargs.get("devices", "auto"), kwargs.get("device"), kwargs.get("context"))
            if kind == "auto":
                ids = _search.auto_devices(self.t, self.y, test_statistic_periods, table, params)
                kind, what = ("list", ids) if ids else ("one", None)
                if kind == "list":
                    kwargs = dict(kwargs, devices=ids)
            if kind == "one":
                fused = _search.fused_power(self.t, self.y, self.dy, test_statistic_periods, table, params,
                                            self.oversampling_factor, context=kwargs.get("context"), device=what)
        if fused is not None:
            ctx_used, summary, chi2, test_statistic_rows, test_statistic_depths, SR, power_raw, power = fused
            kwargs = dict(kwargs, context=ctx_used, device=None)
            if self.verbose:
                print("GPU search on " + ctx_used.name + ": " + str(len(chi2)) + " periods")
        else:
            # (resolved inside search_periods, the one function that knows the HIP library; `used` brings back the context the
            # rest of power() -- spectra, final T0 fit -- runs on: the single device's, or the group's first)
            used = {}
            chi2, test_statistic_rows, test_statistic_depths = _search.search_periods(
                self.t, self.y, self.dy, test_statistic_periods, table,
                context=kwargs.get("context"), device=kwargs.get("device"), devices=kwargs.get("devices", "auto"),
                verbose=self.verbose, used=used, **params)
            if used.get("context") is not None:
                kwargs = dict(kwargs, context=used["context"], device=None)

        idx_best = numpy.argmin(chi2)
        best_row = test_statistic_rows[idx_best]
        duration = lc_cache_overview["duration"][best_row]
        # not rounded up to even here, unlike the search (main.py:201)
        maxwidth_in_samples = int(numpy.max(durations) * numpy.size(self.t))

        no_transits_were_fit = numpy.max(chi2) == numpy.min(chi2)
        if no_transits_were_fit:
            warnings.warn('No transit were fit. Try smaller "transit_depth_min"')

        chi2red = chi2 / (len(self.t) - 4)  # 4 degrees of freedom (main.py:210-212)
        chi2_min = numpy.min(chi2)
        chi2red_min = numpy.min(chi2red)

        if no_transits_were_fit:
            return self._results_without_fit(test_statistic_periods, chi2, chi2red, chi2_min,
                                             chi2red_min)

        if fused is not None:
            SDE_raw, SDE = float(summary["SDE_raw"]), float(summary["SDE"])
        else:
            SR, power_raw, power, SDE_raw, SDE = _search.spectra(chi2, self.oversampling_factor, resident=True,
                                                                 context=kwargs.get("context"), device=kwargs.get("device"))
        # period and depth come from the detrended power peak, the template row from
        # the chi^2 minimum (main.py:198-200 vs 270-272)
        index_highest_power = numpy.argmax(power)
        period = test_statistic_periods[index_highest_power]
        depth = test_statistic_depths[index_highest_power]
        ctx, dev = kwargs.get("context"), kwargs.get("device")
        if fused is not None:
            if self.verbose:
                print("Searching for best T0 for period", format(period, ".5f"), "days")
            T0 = float(summary["T0"])   # (tls_power_prep + tls_t0fit_kernel + tls_first_min: stats.py:135-204 on the device)
        else:
            T0 = final_T0_fit(signal=lc_arr[best_row], depth=depth, t=self.t, y=self.y, dy=self.dy,
                              period=period, T0_fit_margin=self.T0_fit_margin,
                              show_progress_bar=self.show_progress_bar, verbose=self.verbose,
                              residuals_fn=lambda *a: _search.t0_fit_residuals(*a, context=ctx, device=dev))
        transit_times = all_transit_times(T0, self.t, period)
        fill_factor = calculate_fill_factor(self.t)   # (once: the model's fill_half below wants it too)
        transit_duration_in_days = calculate_transit_duration_in_days(
            self.t, period, transit_times, duration, fill_factor=fill_factor)

        phases = fold(self.t, period, T0=T0 + period / 2)
        sort_index = numpy.argsort(phases)
        folded_phase = phases[sort_index]
        folded_y = self.y[sort_index]
        folded_dy = self.dy[sort_index]
        n = numpy.size(self.t)
        # model phase is shifted by half a cadence: mid-transit at phase 0.5
        model_folded_phase = numpy.linspace(0 + 1 / n / 2, 1 + 1 / n / 2, n)

        fill_half = 1 - ((1 - fill_factor) * 0.5)
        stretch = calculate_stretch(self.t, period, transit_times)
        internal_samples = (int(len(self.y) / len(transit_times))
                            * C.OVERSAMPLE_MODEL_LIGHT_CURVE)
        shape = dict(per=self.per, rp=self.rp, a=self.a, inc=self.inc, ecc=self.ecc, w=self.w,
                     u=self.u, limb_dark=self.limb_dark)
        model_folded_model = fractional_transit(
            duration=duration * maxwidth_in_samples * fill_half,
            maxwidth=maxwidth_in_samples / stretch, depth=1 - depth,
            samples=int(len(self.t / len(transit_times))),  # == len(t), main.py:316
            **shape)
        model_transit_single = fractional_transit(
            duration=(duration * maxwidth_in_samples),
            maxwidth=maxwidth_in_samples / stretch, depth=1 - depth,
            samples=internal_samples, **shape)
        model_lightcurve_model, model_lightcurve_time = model_lightcurve(
            transit_times, period, self.t, model_transit_single)

        # (the in-transit flux of every epoch and the out-of-transit flux feed several statistics)
        chunks = _intransit_fluxes(self.t, self.y, transit_times, transit_duration_in_days)
        flux_ootr = self.y[~transit_mask(self.t, period, 2 * duration, T0)]
        (depth_mean_odd, depth_mean_even, depth_mean_odd_std, depth_mean_even_std,
         all_flux_intransit_odd, all_flux_intransit_even, per_transit_count, transit_depths,
         transit_depths_uncertainties) = intransit_stats(
            self.t, self.y, transit_times, transit_duration_in_days, chunks=chunks)
        all_flux_intransit = numpy.concatenate([all_flux_intransit_odd, all_flux_intransit_even])
        std_ootr = numpy.std(flux_ootr)   # (once: the per-epoch SNR and the overall one share it)
        snr_per_transit, snr_pink_per_transit = snr_stats(
            t=self.t, y=self.y, period=period, duration=duration, T0=T0,
            transit_times=transit_times, transit_duration_in_days=transit_duration_in_days,
            per_transit_count=per_transit_count, chunks=chunks, flux_ootr=flux_ootr,
            mean_flux=transit_depths, std_ootr=std_ootr,
            pink_noise_fn=((lambda d, w: _search.pink_noise(d, w, context=ctx, device=dev))
                           if getattr(_search.search_periods, "_tls_amd_product", False) else None))
        depth_mean = numpy.mean(all_flux_intransit)
        depth_mean_std = numpy.std(all_flux_intransit) / numpy.sum(per_transit_count) ** (0.5)
        snr = ((1 - depth_mean) / std_ootr) * len(all_flux_intransit) ** (0.5)
        rp_rs = rp_rs_from_depth(depth=1 - depth, law=self.limb_dark, params=self.u)

        in_transit_count, after_transit_count, before_transit_count = count_stats(
            self.t, self.y, transit_times, transit_duration_in_days)

        odd_even_difference = abs(depth_mean_odd - depth_mean_even)
        odd_even_std_sum = depth_mean_odd_std + depth_mean_even_std
        odd_even_mismatch = odd_even_difference / odd_even_std_sum

        transit_count = len(transit_times)
        empty_transit_count = numpy.count_nonzero(per_transit_count == 0)
        distinct_transit_count = transit_count - empty_transit_count
        if empty_transit_count / transit_count >= 0.33:
            warnings.warn(str(empty_transit_count) + " of " + str(transit_count)
                          + " transits without data. The true period may be twice the"
                          " given period.")

        return transitleastsquaresresults(
            SDE, SDE_raw, chi2_min, chi2red_min, period,
            period_uncertainty(test_statistic_periods, power), T0, transit_duration_in_days,
            depth, (depth_mean, depth_mean_std), (depth_mean_even, depth_mean_even_std),
            (depth_mean_odd, depth_mean_odd_std), transit_depths, transit_depths_uncertainties,
            rp_rs, snr, snr_per_transit, snr_pink_per_transit, odd_even_mismatch, transit_times,
            per_transit_count, transit_count, distinct_transit_count, empty_transit_count,
            FAP(SDE), in_transit_count, after_transit_count, before_transit_count,
            test_statistic_periods, power, power_raw, SR, chi2, chi2red, model_lightcurve_time,
            model_lightcurve_model, model_folded_phase, folded_y, folded_dy, folded_phase,
            model_folded_model)

    def _results_without_fit(self, periods, chi2, chi2red, chi2_min, chi2red_min):
        """Results when no trial passed transit_depth_min (main.py:216-267): flat
        spectra, SDE 0, depth 1, everything else NaN."""
        zeros = numpy.zeros(len(chi2))
        SDE = 0
        return transitleastsquaresresults(
            SDE, 0, chi2_min, chi2red_min, _NAN, period_uncertainty(periods, zeros), 0, _NAN, 1,
            (_NAN, _NAN), (_NAN, _NAN), (_NAN, _NAN), _NAN, _NAN, _NAN, _NAN, _NAN, _NAN, _NAN,
            _NAN, _NAN, _NAN, _NAN, _NAN, FAP(SDE), _NAN, _NAN, _NAN, periods, zeros,
            numpy.zeros(len(chi2)), 0, chi2, chi2red, _NAN, _NAN, _NAN, _NAN, _NAN, _NAN, _NAN)

