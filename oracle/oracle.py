"""ctypes binding of oracle/libtls_oracle.so (TEST INFRASTRUCTURE, see __init__)."""
import ctypes
import os
import subprocess

import numpy

_HERE = os.path.dirname(os.path.abspath(__file__))
_F8 = numpy.ctypeslib.ndpointer(dtype=numpy.float64, flags="C_CONTIGUOUS")
_I8 = numpy.ctypeslib.ndpointer(dtype=numpy.int64, flags="C_CONTIGUOUS")


def build(fast=False, force=False):
    """Compile the oracle with gcc (idempotent). Returns the .so path."""
    name = "libtls_oracle_fast.so" if fast else "libtls_oracle.so"
    path = os.path.join(_HERE, name)
    src = os.path.join(_HERE, "tls_oracle.c")
    if force or not os.path.exists(path) or os.path.getmtime(path) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "fast" if fast else "all"],
                              stdout=subprocess.DEVNULL)
    return path


def usable_cores():
    """Host cores this process may actually use: the affinity mask, capped by the cgroup CPU
    quota (a container that sees 256 CPUs may be throttled to 16 CPUs' worth of time, and 256
    OpenMP threads under such a quota run slower than 16)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    quota = None
    try:
        with open("/sys/fs/cgroup/cpu.max") as fh:            # cgroup v2: "<quota|max> <period>"
            q, per = fh.read().split()[:2]
            if q != "max":
                quota = int(q) / int(per)
    except (OSError, ValueError):
        try:                                                  # cgroup v1
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as fq, \
                    open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as fp:
                q, per = int(fq.read()), int(fp.read())
                if q > 0 and per > 0:
                    quota = q / per
        except (OSError, ValueError):
            pass
    if quota is not None:
        n = min(n, max(1, int(quota + 0.999)))
    return max(1, n)


class OracleLibrary(object):
    def __init__(self, fast=False):
        self.lib = ctypes.CDLL(build(fast=fast))
        f = self.lib.tls_oracle_search
        f.restype = ctypes.c_int
        f.argtypes = [_F8, _F8, _F8, ctypes.c_int64, _F8, ctypes.c_int64,
                      _F8, _I8, _I8, _I8, _F8, ctypes.c_int64,
                      ctypes.c_double, ctypes.c_double, ctypes.c_double, ctypes.c_double,
                      ctypes.c_double, ctypes.c_double,
                      _F8, _I8, _F8, _I8, ctypes.c_int]
        self.lib.tls_oracle_t14.restype = ctypes.c_double
        self.lib.tls_oracle_t14.argtypes = [ctypes.c_double, ctypes.c_double, ctypes.c_double,
                                            ctypes.c_int]
        self.lib.tls_oracle_fold_sort.restype = None
        self.lib.tls_oracle_fold_sort.argtypes = [_F8, ctypes.c_int64, ctypes.c_double, _F8, _I8]
        self.lib.tls_oracle_final_t0_fit.restype = ctypes.c_int64
        self.lib.tls_oracle_final_t0_fit.argtypes = [_F8, ctypes.c_int64, ctypes.c_double, _F8, _F8, ctypes.c_int64,
                                                     ctypes.c_double, ctypes.c_double, _F8, _F8, _F8, ctypes.c_int]
        self.lib.tls_oracle_t0_residuals.restype = ctypes.c_int
        self.lib.tls_oracle_t0_residuals.argtypes = [_F8, _F8, ctypes.c_int64, ctypes.c_double, _F8, ctypes.c_int64,
                                                     _F8, ctypes.c_int64, ctypes.c_int64, _F8, ctypes.c_int]
        self.lib.tls_oracle_spectra.restype = ctypes.c_int
        self.lib.tls_oracle_spectra.argtypes = [_F8, ctypes.c_int64, ctypes.c_int64, _F8, _F8, _F8, _F8]

    def search(self, t, y, dy, periods, table, transit_depth_min, R_star_min, R_star_max,
               M_star_min, M_star_max, T0_fit_margin, n_threads=0):
        """table: tls_amd.template.TemplateTable (or anything with values/offset/
        length/width/overshoot).  Returns chi2, row, depth, counters (len(periods))."""
        c = lambda a, d: numpy.ascontiguousarray(a, dtype=d)
        t, y, dy = c(t, numpy.float64), c(y, numpy.float64), c(dy, numpy.float64)
        periods = c(periods, numpy.float64)
        n_p = len(periods)
        chi2 = numpy.empty(n_p, dtype=numpy.float64)
        row = numpy.empty(n_p, dtype=numpy.int64)
        depth = numpy.empty(n_p, dtype=numpy.float64)
        counters = numpy.zeros(3, dtype=numpy.int64)
        rc = self.lib.tls_oracle_search(
            t, y, dy, len(t), periods, n_p,
            c(table.values, numpy.float64), c(table.offset, numpy.int64),
            c(table.length, numpy.int64), c(table.width, numpy.int64),
            c(table.overshoot, numpy.float64), len(table.width),
            transit_depth_min, R_star_min, R_star_max, M_star_min, M_star_max, T0_fit_margin,
            chi2, row, depth, counters, int(n_threads) if int(n_threads) > 0 else usable_cores())
        if rc != 0:
            raise RuntimeError("tls_oracle_search failed with code %d" % rc)
        return chi2, row, depth, counters

    def final_t0_fit(self, signal, depth, t, y, period, T0_fit_margin, n_threads=0):
        """stats.py:135-204: returns (T0, epochs, residuals) -- the trial grid and the residual of
        every trial epoch besides the reference's return value."""
        c = lambda a: numpy.ascontiguousarray(a, dtype=numpy.float64)
        signal, t, y = c(signal), c(t), c(y)
        n = len(t)
        T0 = numpy.zeros(1)
        epochs, res = numpy.zeros(max(n, 1)), numpy.zeros(max(n, 1))
        points = self.lib.tls_oracle_final_t0_fit(signal, len(signal), float(depth), t, y, n, float(period),
                                                  float(T0_fit_margin), T0, epochs, res,
                                                  int(n_threads) if int(n_threads) > 0 else usable_cores())
        if points < 0:
            raise RuntimeError("tls_oracle_final_t0_fit: bad arguments")
        return float(T0[0]), epochs[:points].copy(), res[:points].copy()

    def t0_residuals(self, t, y, period, signal, epochs, roll, n_threads=0):
        """Loop body of stats.py:178-195 for every trial epoch (signal already depth-scaled)."""
        c = lambda a: numpy.ascontiguousarray(a, dtype=numpy.float64)
        t, y, signal, epochs = c(t), c(y), c(signal), c(epochs)
        out = numpy.empty(len(epochs))
        rc = self.lib.tls_oracle_t0_residuals(t, y, len(t), float(period), signal, len(signal), epochs, len(epochs),
                                              int(roll), out, int(n_threads) if int(n_threads) > 0 else usable_cores())
        if rc != 0:
            raise RuntimeError("tls_oracle_t0_residuals: bad arguments")
        return out

    def spectra(self, chi2, kernel):
        """stats.py:105-132: SR, power_raw, power, SDE_raw, SDE for an integer median kernel."""
        chi2 = numpy.ascontiguousarray(chi2, dtype=numpy.float64)
        n = len(chi2)
        SR, praw, power, sde = numpy.empty(n), numpy.empty(n), numpy.empty(n), numpy.empty(2)
        if self.lib.tls_oracle_spectra(chi2, n, int(kernel), SR, praw, power, sde) != 0:
            raise RuntimeError("tls_oracle_spectra: bad arguments")
        return SR, praw, power, float(sde[0]), float(sde[1])

    def t14(self, R_s, M_s, P, small):
        return self.lib.tls_oracle_t14(R_s, M_s, P, 1 if small else 0)

    def fold_sort(self, t, period):
        t = numpy.ascontiguousarray(t, dtype=numpy.float64)
        ph = numpy.empty(len(t))
        idx = numpy.empty(len(t), dtype=numpy.int64)
        self.lib.tls_oracle_fold_sort(t, len(t), period, ph, idx)
        return ph, idx


_default = {}


def search(*args, **kwargs):
    """Module-level convenience: search with the strict-IEEE oracle build."""
    fast = kwargs.pop("fast", False)
    if fast not in _default:
        _default[fast] = OracleLibrary(fast=fast)
    return _default[fast].search(*args, **kwargs)
