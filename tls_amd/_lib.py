"""ctypes binding of libtls_amd.so -- the C ABI declared in include/tls_amd.h.

Thin by design: numpy arrays in, numpy arrays out, every non-zero return code
raised as RuntimeError with the library's message.  No torch, no fallback: if the
shared library is missing, or no GPU is usable, the error surfaces here.
"""
import ctypes
import os
import weakref

import numpy

_HERE = os.path.dirname(os.path.abspath(__file__))
# Developer switch for A/B timing of two builds of the same source (tls_amd/csrc/Makefile `variant`, `debug`): another
# library is loaded only when TLS_AMD_DEBUG=1 says this is a developer session AND TLS_AMD_LIB names it -- a stray
# TLS_AMD_LIB in a user's environment does not redirect the product.
LIB_PATH = os.path.join(_HERE, "libtls_amd.so")
if os.environ.get("TLS_AMD_DEBUG") == "1" and os.environ.get("TLS_AMD_LIB"):
    LIB_PATH = os.environ["TLS_AMD_LIB"]
ABI_VERSION = 5   # include/tls_amd.h TLS_AMD_ABI_VERSION: checked against the library at load time

# every symbol include/tls_amd.h declares (tests check the export list against the header)
SYMBOLS = (
    "tls_device_count", "tls_ctx_create", "tls_ctx_destroy", "tls_last_error", "tls_version", "tls_abi_version",
    "tls_device_name", "tls_get_options", "tls_set_options", "tls_debug_set_switch", "tls_debug_get_switches", "tls_search", "tls_search_batch", "tls_power_batch", "tls_prepare", "tls_update_flux", "tls_execute",
    "tls_synchronize", "tls_fetch", "tls_execute_timed", "tls_plan_info", "tls_last_kernel", "tls_grid_cells", "tls_period_costs", "tls_t0_fit", "tls_pink_noise", "tls_spectra", "tls_kernel_timing", "tls_debug_phase_cycles", "tls_debug_cumsum", "tls_debug_folded", "tls_debug_prefix", "tls_debug_check_counts", "tls_debug_poison_lds", "tls_debug_period_cycles", "tls_debug_batch_group_ms",
    "tls_comm_unique_id", "tls_comm_init", "tls_comm_destroy", "tls_comm_info", "tls_comm_allgather_results", "tls_comm_allgather_device", "tls_comm_fetch_gathered",
    "tls_comm_stage_results", "tls_comm_allgather_staged", "tls_comm_fetch_staged",
    "tls_comm_barrier", "tls_comm_max",
)

_c_double_p = ctypes.POINTER(ctypes.c_double)
_c_int64_p = ctypes.POINTER(ctypes.c_int64)


class _Template(ctypes.Structure):
    _fields_ = [("values", _c_double_p), ("offset", _c_int64_p), ("length", _c_int64_p),
                ("width", _c_int64_p), ("overshoot", _c_double_p), ("n_rows", ctypes.c_int64)]


class _Params(ctypes.Structure):
    _fields_ = [("transit_depth_min", ctypes.c_double), ("R_star_min", ctypes.c_double),
                ("R_star_max", ctypes.c_double), ("M_star_min", ctypes.c_double),
                ("M_star_max", ctypes.c_double), ("T0_fit_margin", ctypes.c_double)]


class Options(ctypes.Structure):
    """tls_options: the two switches of a context a caller may set (include/tls_amd.h).  -1 = the library decides."""
    _fields_ = [("exact_prefix", ctypes.c_int32), ("slim", ctypes.c_int32)]

    NAMES = ("exact_prefix", "slim")


# every switch tls_debug_set_switch knows (the two public ones included): Context.set_options(**switches) takes them all
SWITCH_NAMES = ("exact_prefix", "slim", "prune", "screen32", "no_screen", "fast_slab", "x_staged", "split", "split_batch", "sort2",
                "threads", "blocks", "plan_threads", "t0_rot", "prune_min_live", "band_max")


def switches_text(switches):
    """"name=value,..." (what tls_period_costs and tls_debug_get_switches speak) from a dict; None values are left out."""
    unknown = set(switches or {}) - set(SWITCH_NAMES)
    if unknown:
        raise TypeError("unknown switches %s" % sorted(unknown))
    return ",".join("%s=%.17g" % (k, float(v)) for k, v in sorted((switches or {}).items()) if v is not None)


class PowerSummary(ctypes.Structure):
    """tls_power_summary: what main.py:198-283 derives from the search results of one light curve."""
    _fields_ = [("SDE", ctypes.c_double), ("SDE_raw", ctypes.c_double), ("chi2_min", ctypes.c_double),
                ("period", ctypes.c_double), ("T0", ctypes.c_double), ("depth", ctypes.c_double),
                ("index_best", ctypes.c_int64), ("index_power", ctypes.c_int64), ("best_row", ctypes.c_int64),
                ("no_fit", ctypes.c_int64)]


POWER_SUMMARY_DTYPE = numpy.dtype([("SDE", "f8"), ("SDE_raw", "f8"), ("chi2_min", "f8"), ("period", "f8"), ("T0", "f8"),
                                   ("depth", "f8"), ("index_best", "i8"), ("index_power", "i8"), ("best_row", "i8"),
                                   ("no_fit", "i8")])


class Counters(ctypes.Structure):
    _fields_ = [("grid_cells", ctypes.c_int64), ("evaluated_cells", ctypes.c_int64),
                ("inner_steps", ctypes.c_int64), ("pd_pairs", ctypes.c_int64),
                ("issued_fma", ctypes.c_int64)]

    def as_dict(self):
        return {k: int(getattr(self, k)) for k, _ in self._fields_}


_lib = None


def load():
    """Load libtls_amd.so once; raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            "tls_amd: %s is missing -- build it with `python -c 'import __graft_entry__ as g; "
            "g.build()'` or `make -C tls_amd/csrc`; there is no CPU fallback" % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    vp, ci, i64, dbl = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_double
    lib.tls_abi_version.restype = ci
    if lib.tls_abi_version() != ABI_VERSION:
        raise RuntimeError("tls_amd: %s implements ABI version %d, this binding expects %d (struct layouts differ): "
                           "rebuild with `make -C tls_amd/csrc`" % (LIB_PATH, lib.tls_abi_version(), ABI_VERSION))
    lib.tls_device_count.restype = ci
    lib.tls_ctx_create.restype = vp
    lib.tls_ctx_create.argtypes = [ci]
    lib.tls_ctx_destroy.restype = None
    lib.tls_ctx_destroy.argtypes = [vp]
    lib.tls_last_error.restype = ctypes.c_char_p
    lib.tls_last_error.argtypes = [vp]
    lib.tls_version.restype = ctypes.c_char_p
    lib.tls_device_name.restype = ctypes.c_char_p
    lib.tls_device_name.argtypes = [vp]
    lib.tls_get_options.restype = ci
    lib.tls_get_options.argtypes = [vp, ctypes.POINTER(Options)]
    lib.tls_set_options.restype = ci
    lib.tls_set_options.argtypes = [vp, ctypes.POINTER(Options)]
    lib.tls_debug_set_switch.restype = ci
    lib.tls_debug_set_switch.argtypes = [vp, ctypes.c_char_p, dbl]
    lib.tls_debug_get_switches.restype = ci
    lib.tls_debug_get_switches.argtypes = [vp, ctypes.c_char_p, i64]
    tp, pp = ctypes.POINTER(_Template), ctypes.POINTER(_Params)
    cp = ctypes.POINTER(Counters)
    lib.tls_search.restype = ci
    lib.tls_search.argtypes = [vp, _c_double_p, _c_double_p, _c_double_p, i64, _c_double_p, i64,
                               tp, pp, _c_double_p, _c_int64_p, _c_double_p, cp]
    lib.tls_search_batch.restype = ci
    lib.tls_search_batch.argtypes = [vp, _c_double_p, _c_double_p, _c_double_p, i64, i64, _c_double_p, i64,
                                     tp, pp, _c_double_p, _c_int64_p, _c_double_p]
    lib.tls_power_batch.restype = ci
    lib.tls_power_batch.argtypes = [vp, _c_double_p, _c_double_p, _c_double_p, i64, i64, _c_double_p, i64,
                                    tp, pp, i64, ctypes.c_void_p, _c_double_p, _c_int64_p, _c_double_p, _c_double_p,
                                    _c_double_p, _c_double_p]
    lib.tls_prepare.restype = ci
    lib.tls_prepare.argtypes = [vp, _c_double_p, _c_double_p, _c_double_p, i64, _c_double_p, i64,
                                tp, pp]
    lib.tls_update_flux.restype = ci
    lib.tls_update_flux.argtypes = [vp, _c_double_p, _c_double_p]
    lib.tls_execute.restype = ci
    lib.tls_execute.argtypes = [vp, ci]
    lib.tls_synchronize.restype = ci
    lib.tls_synchronize.argtypes = [vp]
    lib.tls_fetch.restype = ci
    lib.tls_fetch.argtypes = [vp, _c_double_p, _c_int64_p, _c_double_p, cp]
    lib.tls_execute_timed.restype = ci
    lib.tls_execute_timed.argtypes = [vp, ci, _c_double_p]
    lib.tls_last_kernel.restype = ctypes.c_char_p
    lib.tls_last_kernel.argtypes = [vp]
    lib.tls_plan_info.restype = ci
    lib.tls_plan_info.argtypes = [vp, cp, _c_int64_p, _c_int64_p, _c_int64_p]
    lib.tls_kernel_timing.restype = ci
    lib.tls_kernel_timing.argtypes = [vp, ci, _c_double_p, _c_int64_p]
    lib.tls_debug_phase_cycles.restype = ci
    lib.tls_debug_phase_cycles.argtypes = [vp, ctypes.POINTER(ctypes.c_uint64), ci]
    lib.tls_debug_poison_lds.restype = ci
    lib.tls_debug_poison_lds.argtypes = [vp, ctypes.c_uint32]
    lib.tls_debug_check_counts.restype = ci
    lib.tls_debug_check_counts.argtypes = [vp, ctypes.POINTER(ctypes.c_uint64), ci]
    lib.tls_t0_fit.restype = ci
    lib.tls_t0_fit.argtypes = [vp, _c_double_p, _c_double_p, i64, dbl, _c_double_p, i64, _c_double_p,
                               i64, i64, _c_double_p]
    lib.tls_pink_noise.restype = ci
    lib.tls_pink_noise.argtypes = [vp, _c_double_p, i64, i64, ctypes.c_double, _c_double_p]
    lib.tls_spectra.restype = ci
    lib.tls_spectra.argtypes = [vp, _c_double_p, i64, i64, _c_double_p, _c_double_p, _c_double_p, _c_double_p]
    lib.tls_debug_folded.restype = ci
    lib.tls_debug_folded.argtypes = [vp, _c_double_p, i64]
    lib.tls_debug_prefix.restype = ci
    lib.tls_debug_prefix.argtypes = [vp, _c_double_p, i64, ctypes.POINTER(ctypes.c_int64)]
    lib.tls_debug_period_cycles.restype = ci
    lib.tls_debug_period_cycles.argtypes = [vp, ctypes.POINTER(ctypes.c_uint64), i64]
    lib.tls_debug_batch_group_ms.restype = ci
    lib.tls_debug_batch_group_ms.argtypes = [vp, _c_double_p, i64]
    lib.tls_debug_cumsum.restype = ci
    lib.tls_debug_cumsum.argtypes = [vp, _c_double_p, i64, _c_double_p, ci]
    lib.tls_grid_cells.restype = ci
    lib.tls_grid_cells.argtypes = [_c_double_p, i64, _c_double_p, i64, tp, pp, _c_int64_p]
    lib.tls_period_costs.restype = ci
    lib.tls_period_costs.argtypes = [_c_double_p, i64, _c_double_p, i64, tp, pp, dbl, _c_int64_p, _c_double_p, _c_double_p,
                                     _c_int64_p, ctypes.c_char_p]
    lib.tls_comm_unique_id.restype = ci
    lib.tls_comm_unique_id.argtypes = [ctypes.c_char_p]
    lib.tls_comm_init.restype = ci
    lib.tls_comm_init.argtypes = [vp, ci, ci, ctypes.c_char_p]
    lib.tls_comm_destroy.restype = ci
    lib.tls_comm_destroy.argtypes = [vp]
    lib.tls_comm_info.restype = ci
    lib.tls_comm_info.argtypes = [vp, ctypes.POINTER(ci), ctypes.POINTER(ci), ctypes.POINTER(ci)]
    lib.tls_comm_allgather_results.restype = ci
    lib.tls_comm_allgather_results.argtypes = [vp, i64, _c_double_p, _c_int64_p, _c_double_p]
    lib.tls_comm_allgather_device.restype = ci
    lib.tls_comm_allgather_device.argtypes = [vp, i64]
    lib.tls_comm_fetch_gathered.restype = ci
    lib.tls_comm_fetch_gathered.argtypes = [vp, i64, _c_double_p, _c_int64_p, _c_double_p]
    lib.tls_comm_stage_results.restype = ci
    lib.tls_comm_stage_results.argtypes = [vp, i64, i64, i64]
    lib.tls_comm_allgather_staged.restype = ci
    lib.tls_comm_allgather_staged.argtypes = [vp, i64, i64]
    lib.tls_comm_fetch_staged.restype = ci
    lib.tls_comm_fetch_staged.argtypes = [vp, i64, i64, i64, _c_double_p, _c_int64_p, _c_double_p]
    lib.tls_comm_barrier.restype = ci
    lib.tls_comm_barrier.argtypes = [vp]
    lib.tls_comm_max.restype = ci
    lib.tls_comm_max.argtypes = [vp, _c_double_p]
    _lib = lib
    return lib


def _f8(a):
    return numpy.ascontiguousarray(a, dtype=numpy.float64)


def _i8(a):
    return numpy.ascontiguousarray(a, dtype=numpy.int64)


def _dp(a):
    return a.ctypes.data_as(_c_double_p)


def _ip(a):
    return a.ctypes.data_as(_c_int64_p)


class Context(object):
    """One GPU + one HIP stream (tls_ctx).  Not re-entrant."""

    def __init__(self, device=0):
        self._lib = load()
        self._h = self._lib.tls_ctx_create(int(device))
        if not self._h:
            raise RuntimeError("tls_amd: cannot create a GPU context: "
                               + self._lib.tls_last_error(None).decode())
        self.device = int(device)
        self._n_periods = 0
        self._resident_chi2 = None   # (weak reference to the chi2 array the last fetch returned, generation): see holds
        self._generation = 0         # bumped by every call that launches a search or touches its inputs / result buffers

    # -- plumbing
    def close(self):
        if getattr(self, "_h", None):
            self._lib.tls_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc != 0:
            raise RuntimeError("tls_amd error %d: %s"
                               % (rc, self._lib.tls_last_error(self._h).decode()))

    @property
    def name(self):
        return self._lib.tls_device_name(self._h).decode()

    # -- switches (tls_options): the environment is read once per process, when the library is first used; a context
    #    starts with those values and changes them only through these calls
    def get_options(self):
        """Every switch of the context by name (the public tls_options and the developer switches): -1 = the library decides."""
        buf = ctypes.create_string_buffer(1024)
        rc = self._lib.tls_debug_get_switches(self._h, buf, len(buf))
        if rc != 0:
            raise RuntimeError("tls_amd error %d: tls_debug_get_switches" % rc)
        out = {}
        for item in buf.value.decode().split(","):
            k, _, v = item.partition("=")
            out[k] = float(v) if k == "band_max" else int(float(v))
        return out

    def set_options(self, **switches):
        """Set the named switches (None: back to "the library decides"); the others keep their values.  Drops a
        prepared plan: the next prepare()/search() plans again.  exact_prefix and slim go through the public
        tls_set_options, the developer switches through tls_debug_set_switch."""
        for k in switches:
            if k not in SWITCH_NAMES:
                raise TypeError("unknown switch %r (known: %s)" % (k, ", ".join(SWITCH_NAMES)))
        self._invalidate_results()
        public = {k: v for k, v in switches.items() if k in Options.NAMES}
        if public:
            o = Options()
            self._check(self._lib.tls_get_options(self._h, ctypes.byref(o)))
            for k, v in public.items():
                setattr(o, k, -1 if v is None else int(v))
            self._check(self._lib.tls_set_options(self._h, ctypes.byref(o)))
        for k, v in switches.items():
            if k not in Options.NAMES:
                self._check(self._lib.tls_debug_set_switch(self._h, k.encode(), -1.0 if v is None else float(v)))

    def reset_options(self):
        """Every switch back to "the library decides" (NOT to the process environment's values)."""
        self.set_options(**{k: None for k in SWITCH_NAMES})

    @staticmethod
    def _pack(table, params):
        arrays = (_f8(table.values), _i8(table.offset), _i8(table.length), _i8(table.width),
                  _f8(table.overshoot))
        tm = _Template(_dp(arrays[0]), _ip(arrays[1]), _ip(arrays[2]), _ip(arrays[3]),
                       _dp(arrays[4]), len(arrays[3]))
        pr = _Params(*[float(params[k]) for k in (
            "transit_depth_min", "R_star_min", "R_star_max", "M_star_min", "M_star_max",
            "T0_fit_margin")])
        return arrays, tm, pr

    # -- one-shot
    def search(self, t, y, dy, periods, table, params, count_work=False):
        """chi2, row, depth (and counters dict) for every period, in `periods` order.

        Consecutive searches on the same time stamps, period list, template table and parameters
        (a survey, or repeated power() calls) reuse the prepared plan: tls_prepare recognises them
        (byte comparison inside the library) and only replaces the flux and the weights."""
        self.prepare(t, y, dy, periods, table, params)
        self.execute(count_work=count_work)
        return self.fetch(with_counters=True)

    # -- staged
    def search_batch(self, t, y_batch, dy_batch, periods, table, params):
        """Survey mode: chi2, row, depth of shape [n_curves, n_periods] for light curves that share
        t, the grids and the template table (tls_search_batch)."""
        t, periods = _f8(t), _f8(periods)
        y_batch = numpy.ascontiguousarray(y_batch, dtype=numpy.float64)
        dy_batch = numpy.ascontiguousarray(dy_batch, dtype=numpy.float64)
        if y_batch.ndim != 2 or y_batch.shape != dy_batch.shape or y_batch.shape[1] != len(t):
            raise ValueError("y_batch and dy_batch must both have shape [n_curves, len(t)]")
        arrays, tm, pr = self._pack(table, params)
        n_c, n_p = y_batch.shape[0], len(periods)
        chi2 = numpy.empty((n_c, n_p), dtype=numpy.float64)
        row = numpy.empty((n_c, n_p), dtype=numpy.int64)
        depth = numpy.empty((n_c, n_p), dtype=numpy.float64)
        self._invalidate_results()
        self._check(self._lib.tls_search_batch(self._h, _dp(t), _dp(y_batch), _dp(dy_batch), len(t), n_c,
                                               _dp(periods), n_p, ctypes.byref(tm), ctypes.byref(pr),
                                               _dp(chi2), _ip(row), _dp(depth)))
        self._n_periods = n_p
        return chi2, row, depth

    def power_batch(self, t, y_batch, dy_batch, periods, table, params, median_kernel, with_arrays=False,
                    with_power=False, with_spectra=False):
        """Survey-mode power(): structured array (POWER_SUMMARY_DTYPE) with one record per light curve -- SDE,
        SDE_raw, chi2_min, period, T0, depth, the argmin/argmax indices, the template row -- from search,
        spectra and final T0 fit on the device (tls_power_batch); optionally the per-period arrays.
        with_spectra: also SR and power_raw (returned as two more arrays behind `power`)."""
        t, periods = _f8(t), _f8(periods)
        y_batch = numpy.ascontiguousarray(y_batch, dtype=numpy.float64)
        dy_batch = numpy.ascontiguousarray(dy_batch, dtype=numpy.float64)
        if y_batch.ndim != 2 or y_batch.shape != dy_batch.shape or y_batch.shape[1] != len(t):
            raise ValueError("y_batch and dy_batch must both have shape [n_curves, len(t)]")
        arrays, tm, pr = self._pack(table, params)
        n_c, n_p = y_batch.shape[0], len(periods)
        summary = numpy.zeros(n_c, dtype=POWER_SUMMARY_DTYPE)
        assert summary.dtype.itemsize == ctypes.sizeof(PowerSummary)
        chi2 = row = depth = power = None
        if with_arrays:
            chi2 = numpy.empty((n_c, n_p), dtype=numpy.float64)
            row = numpy.empty((n_c, n_p), dtype=numpy.int64)
            depth = numpy.empty((n_c, n_p), dtype=numpy.float64)
        SR = power_raw = None
        if with_power:
            power = numpy.empty((n_c, n_p), dtype=numpy.float64)
        if with_spectra:
            SR = numpy.empty((n_c, n_p), dtype=numpy.float64)
            power_raw = numpy.empty((n_c, n_p), dtype=numpy.float64)
        self._invalidate_results()
        self._check(self._lib.tls_power_batch(
            self._h, _dp(t), _dp(y_batch), _dp(dy_batch), len(t), n_c, _dp(periods), n_p, ctypes.byref(tm),
            ctypes.byref(pr), int(median_kernel), summary.ctypes.data_as(ctypes.c_void_p),
            None if chi2 is None else _dp(chi2), None if row is None else _ip(row),
            None if depth is None else _dp(depth), None if power is None else _dp(power),
            None if SR is None else _dp(SR), None if power_raw is None else _dp(power_raw)))
        self._n_periods = n_p
        if with_spectra:
            return summary, chi2, row, depth, power, SR, power_raw
        return summary, chi2, row, depth, power

    def _invalidate_results(self):
        """Every call that launches a search, replaces its inputs or reuses the result buffers: the chi2 array an
        earlier fetch returned is no longer what the device holds (see holds)."""
        self._resident_chi2 = None
        self._generation += 1

    def prepare(self, t, y, dy, periods, table, params):
        self._invalidate_results()
        t, y, dy, periods = _f8(t), _f8(y), _f8(dy), _f8(periods)
        if not (t.ndim == y.ndim == dy.ndim == 1 and len(t) == len(y) == len(dy)):
            raise ValueError("t, y, dy must be 1-dimensional and of equal length")
        arrays, tm, pr = self._pack(table, params)
        self._check(self._lib.tls_prepare(self._h, _dp(t), _dp(y), _dp(dy), len(t), _dp(periods),
                                          len(periods), ctypes.byref(tm), ctypes.byref(pr)))
        self._n_periods = len(periods)

    def update_flux(self, y, dy):
        self._invalidate_results()
        y, dy = _f8(y), _f8(dy)
        self._check(self._lib.tls_update_flux(self._h, _dp(y), _dp(dy)))

    def execute(self, count_work=False, phase_clock=False):
        self._invalidate_results()
        self._check(self._lib.tls_execute(self._h, (1 if count_work else 0) | (2 if phase_clock else 0)))

    def t0_fit_residuals(self, t, y, period, signal, epochs, roll):
        """Residual of the depth-scaled template at every trial epoch (stats.py:178-195)."""
        t, y, signal, epochs = _f8(t), _f8(y), _f8(signal), _f8(epochs)
        out = numpy.empty(len(epochs), dtype=numpy.float64)
        self._check(self._lib.tls_t0_fit(self._h, _dp(t), _dp(y), len(t), float(period), _dp(signal),
                                         len(signal), _dp(epochs), len(epochs), int(roll), _dp(out)))
        return out

    @staticmethod
    def _fingerprint(chi2):
        """Cheap value check of a fetched array (one pass): catches an in-place edit between fetch and spectra."""
        if len(chi2) == 0:
            return (0, 0.0)
        return (len(chi2), float(numpy.sum(chi2)), float(chi2[0]), float(chi2[-1]), float(chi2[len(chi2) // 2]))

    def holds(self, chi2):
        """True if `chi2` is the very array the last fetch of this context returned, no call has launched a search,
        replaced the flux or reused the result buffers since (every such method bumps the context's generation), and
        the array still has the values it was handed out with: the device then holds the same values and tls_spectra
        may read them in place.  Anything else is uploaded."""
        ref = self._resident_chi2
        return (ref is not None and ref[0]() is chi2 and ref[1] == self._generation and len(chi2) == self._n_periods
                and ref[2] == self._fingerprint(chi2))

    def spectra(self, kernel, chi2=None):
        """SR, power_raw, power, SDE_raw, SDE (stats.py:105-132) on the device; chi2=None takes the
        chi^2 of the search that has just finished (still resident in HBM)."""
        n = self._n_periods if chi2 is None else len(chi2)
        block = numpy.empty(3 * n + 2, dtype=numpy.float64)     # one block: one device-to-host copy
        SR, praw, power, sde = block[:n], block[n:2 * n], block[2 * n:3 * n], block[3 * n:]
        c = None if chi2 is None else _f8(chi2)
        self._check(self._lib.tls_spectra(self._h, None if c is None else _dp(c), n, int(kernel), _dp(SR), _dp(praw),
                                          _dp(power), _dp(sde)))
        return SR, praw, power, float(sde[0]), float(sde[1])

    def pink_noise(self, data, width):
        """Mean over all windows of `width` points of std(window) / width ** 0.5 (stats.py:72-77) on the device, with
        the roundings of the reference's loop over numpy.std."""
        d = _f8(data)
        out = numpy.empty(1, dtype=numpy.float64)
        self._check(self._lib.tls_pink_noise(self._h, _dp(d), len(d), int(width), float(int(width) ** 0.5), _dp(out)))
        return float(out[0])

    def debug_cumsum(self, values, threads=512):
        """[0, cumsum(values)] computed by the kernel's exact parallel sequential-order scan."""
        v = _f8(values)
        out = numpy.empty(len(v) + 1, dtype=numpy.float64)
        self._check(self._lib.tls_debug_cumsum(self._h, _dp(v), len(v), _dp(out), int(threads)))
        return out

    def folded(self, n_periods, n):
        """Developer/test entry: the folded flux of every period of the prepared plan, (n_periods, n), as the
        kernel's sort left it."""
        self._invalidate_results()
        out = numpy.empty((int(n_periods), int(n)), dtype=numpy.float64)
        self._check(self._lib.tls_debug_folded(self._h, _dp(out), out.size))
        return out

    def prefix_sums(self, n_periods):
        """Developer/test entry: the prefix sum C[0..M] of the patched folded flux of every period, (n_periods, M + 1)."""
        self._invalidate_results()
        row = ctypes.c_int64(0)
        self._check(self._lib.tls_debug_prefix(self._h, None, 0, ctypes.byref(row)))
        out = numpy.empty((int(n_periods), int(row.value)), dtype=numpy.float64)
        self._check(self._lib.tls_debug_prefix(self._h, _dp(out), out.size, ctypes.byref(row)))
        return out

    def phase_cycles(self):
        """Developer instrumentation: per-phase shader-cycle sums of the last
        execute(phase_clock=True)."""
        arr = (ctypes.c_uint64 * 40)()
        self._check(self._lib.tls_debug_phase_cycles(self._h, arr, 40))
        names = ("fold_count", "scan", "scatter", "rank", "gather_patch", "cumsum", "batch_prefix",
                 "chi2", "e_convert", "predicate_strided", "cumsum_blocks", "cumsum_fallbacks",
                 "tile_staging", "predicate_dense", "cs_A", "cs_B1", "cs_B2", "cs_scan", "screen_split", "screen_resolve",
                 "tile_wait", "chi2_wait", "prune_e2", "prune_bounds", "prune_incumbent", "select_relist",
                 "slab_copy_in", "slab_copy_out", "part_fold", "part_scan", "part_lds", "part_store",
                 "stat_live_units", "stat_kept_units", "stat_singles", "stat_batches", "stat_pruned_periods",
                 "stat_exact_retries", "stat_screen_parked", "stat_screen_valued")
        return dict(zip(names, [int(v) for v in arr]))

    def period_cycles(self):
        """Developer instrumentation: shader cycles per period of the prepared plan (one more search)."""
        self._invalidate_results()
        out = numpy.zeros(self._n_periods, dtype=numpy.uint64)
        self._check(self._lib.tls_debug_period_cycles(self._h, out.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64)),
                                                      len(out)))
        return out

    def batch_group_ms(self, with_wait=False):
        """Host wall time (ms) of every group of 32 light curves of the last power_batch / search_batch call; with_wait: also
        the part of each group's time spent in its one wait for the device (power_batch)."""
        n = self._lib.tls_debug_batch_group_ms(self._h, None, 0)
        out = numpy.zeros(2 * max(n, 0), dtype=numpy.float64)
        if n > 0:
            self._lib.tls_debug_batch_group_ms(self._h, _dp(out), 2 * n)
        return (out[:n], out[n:]) if with_wait else out[:n]

    def check_counts(self):
        """(checked_build, {check name: violations}) -- device-side bound checks of the debug build."""
        arr = (ctypes.c_uint64 * 16)()
        rc = self._lib.tls_debug_check_counts(self._h, arr, 16)
        if rc < 0:
            self._check(rc)
        names = ("lds_carve", "list_capacity", "dot_window", "predicate_read", "sort_window", "work_item",
                 "singles_capacity", "tile_stage", "screen_split")
        return bool(rc), dict(zip(names, [int(v) for v in arr]))

    def poison_lds(self, word=0x7ff80000):
        """Test entry: every CU's LDS filled with `word` (default: fp64 NaNs) on the context's stream."""
        self._check(self._lib.tls_debug_poison_lds(self._h, int(word)))

    def synchronize(self):
        self._check(self._lib.tls_synchronize(self._h))

    def execute_timed(self, reps=1):
        self._invalidate_results()
        ms = ctypes.c_double(0.0)
        self._check(self._lib.tls_execute_timed(self._h, int(reps), ctypes.byref(ms)))
        return ms.value

    def fetch(self, with_counters=False):
        n = self._n_periods
        chi2 = numpy.empty(n, dtype=numpy.float64)
        row = numpy.empty(n, dtype=numpy.int64)
        depth = numpy.empty(n, dtype=numpy.float64)
        c = Counters()
        self._check(self._lib.tls_fetch(self._h, _dp(chi2), _ip(row), _dp(depth), ctypes.byref(c)))
        # exactly this array, at this generation of the context, with these values is what the device still holds
        self._resident_chi2 = (weakref.ref(chi2), self._generation, self._fingerprint(chi2))
        if with_counters:
            return chi2, row, depth, c.as_dict()
        return chi2, row, depth

    def kernel_timing(self, reset=True):
        """(total ms, launches) of the search kernel since the last reset (HIP events)."""
        ms, n = ctypes.c_double(0.0), ctypes.c_int64(0)
        self._check(self._lib.tls_kernel_timing(self._h, 1 if reset else 0, ctypes.byref(ms),
                                                ctypes.byref(n)))
        return ms.value, n.value

    def plan_info(self):
        c = Counters()
        lds, blocks, res = ctypes.c_int64(), ctypes.c_int64(), ctypes.c_int64()
        self._check(self._lib.tls_plan_info(self._h, ctypes.byref(c), ctypes.byref(lds),
                                            ctypes.byref(blocks), ctypes.byref(res)))
        d = c.as_dict()
        d.update(lds_bytes=lds.value, n_blocks=blocks.value, resident=bool(res.value))
        return d

    def last_kernel(self):
        """Name of the search kernel the last execute launched (include/tls_amd.h, tls_last_kernel)."""
        return self._lib.tls_last_kernel(self._h).decode()

    # -- RCCL
    def comm_unique_id(self):
        buf = ctypes.create_string_buffer(128)
        rc = self._lib.tls_comm_unique_id(buf)
        if rc != 0:
            raise RuntimeError("tls_amd: ncclGetUniqueId failed: "
                               + self._lib.tls_last_error(None).decode())
        return buf.raw

    def comm_init(self, n_ranks, rank, unique_id):
        assert len(unique_id) == 128
        self._check(self._lib.tls_comm_init(self._h, int(n_ranks), int(rank), unique_id))

    def comm_destroy(self):
        self._check(self._lib.tls_comm_destroy(self._h))

    def comm_info(self):
        """(ranks, this rank, device) as RCCL reports them for the communicator; (0, -1, -1) without one."""
        n, r, d = ctypes.c_int(0), ctypes.c_int(-1), ctypes.c_int(-1)
        self._check(self._lib.tls_comm_info(self._h, ctypes.byref(n), ctypes.byref(r), ctypes.byref(d)))
        return n.value, r.value, d.value

    def comm_ranks(self):
        return self.comm_info()[0]

    def comm_allgather_results(self, count_per_rank, n_ranks):
        total = int(count_per_rank) * int(n_ranks)
        chi2 = numpy.empty(total, dtype=numpy.float64)
        row = numpy.empty(total, dtype=numpy.int64)
        depth = numpy.empty(total, dtype=numpy.float64)
        self._check(self._lib.tls_comm_allgather_results(self._h, int(count_per_rank), _dp(chi2),
                                                         _ip(row), _dp(depth)))
        return chi2, row, depth

    def comm_allgather_device(self, count_per_rank):
        """Enqueue pack + ncclAllGather behind the search; the batch stays device resident."""
        self._check(self._lib.tls_comm_allgather_device(self._h, int(count_per_rank)))

    def comm_stage_results(self, count_per_rank, slot, n_slots):
        """Survey mode: park the latest results as slot `slot` (device copies, no communication)."""
        self._check(self._lib.tls_comm_stage_results(self._h, int(count_per_rank), int(slot), int(n_slots)))

    def comm_allgather_staged(self, count_per_rank, n_slots):
        """ONE ncclAllGather of all staged slots, enqueued on the search stream."""
        self._check(self._lib.tls_comm_allgather_staged(self._h, int(count_per_rank), int(n_slots)))

    def comm_fetch_staged(self, count_per_rank, n_slots, slot, n_ranks):
        n = int(count_per_rank) * int(n_ranks)
        chi2 = numpy.empty(n, dtype=numpy.float64)
        row = numpy.empty(n, dtype=numpy.int64)
        depth = numpy.empty(n, dtype=numpy.float64)
        self._check(self._lib.tls_comm_fetch_staged(self._h, int(count_per_rank), int(n_slots), int(slot),
                                                    _dp(chi2), _ip(row), _dp(depth)))
        return chi2, row, depth

    def comm_fetch_gathered(self, count_per_rank, n_ranks):
        total = int(count_per_rank) * int(n_ranks)
        chi2 = numpy.empty(total, dtype=numpy.float64)
        row = numpy.empty(total, dtype=numpy.int64)
        depth = numpy.empty(total, dtype=numpy.float64)
        self._check(self._lib.tls_comm_fetch_gathered(self._h, int(count_per_rank), _dp(chi2), _ip(row),
                                                      _dp(depth)))
        return chi2, row, depth

    def comm_barrier(self):
        self._check(self._lib.tls_comm_barrier(self._h))

    def comm_max(self, value):
        v = ctypes.c_double(float(value))
        self._check(self._lib.tls_comm_max(self._h, ctypes.byref(v)))
        return v.value


def grid_cells(t, periods, table, params):
    """Trial cells each period enumerates (host-only planning call, needs no GPU)."""
    lib = load()
    t, periods = _f8(t), _f8(periods)
    arrays, tm, pr = Context._pack(table, params)
    out = numpy.zeros(len(periods), dtype=numpy.int64)
    rc = lib.tls_grid_cells(_dp(t), len(t), _dp(periods), len(periods), ctypes.byref(tm),
                            ctypes.byref(pr), _ip(out))
    if rc != 0:
        raise RuntimeError("tls_amd error %d: %s" % (rc, lib.tls_last_error(None).decode()))
    return out


def period_costs(t, periods, table, params, sigma, with_slots=False, options=None):
    """(trial cells, expected template taps, modelled search time) of every period: what the shard
    boundaries are placed by (host-only planning call, needs no GPU).  options: the switches of the context
    that will search (Context.get_options()) -- the model follows the kernel variant and prefix-sum mode they
    select; None: the process's (its TLS_* environment, read once)."""
    lib = load()
    t, periods = _f8(t), _f8(periods)
    arrays, tm, pr = Context._pack(table, params)
    cells = numpy.zeros(len(periods), dtype=numpy.int64)
    taps = numpy.zeros(len(periods), dtype=numpy.float64)
    time = numpy.zeros(len(periods), dtype=numpy.float64)
    slots = ctypes.c_int64(0)
    opt = None if options is None else switches_text({k: v for k, v in options.items() if v is not None and float(v) >= 0}).encode()
    rc = lib.tls_period_costs(_dp(t), len(t), _dp(periods), len(periods), ctypes.byref(tm), ctypes.byref(pr),
                              float(sigma), _ip(cells), _dp(taps), _dp(time), ctypes.byref(slots), opt)
    if rc != 0:
        raise RuntimeError("tls_amd error %d: %s" % (rc, lib.tls_last_error(None).decode()))
    if with_slots:
        return cells, taps, time, int(slots.value)
    return cells, taps, time


def device_count():
    lib = load()
    n = lib.tls_device_count()
    if n < 0:
        raise RuntimeError("tls_amd: " + lib.tls_last_error(None).decode())
    return n
