"""Developer tool: does the fast path of the exact prefix sum hold, and if not, which check fails."""
import os, sys
import numpy
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tls_amd import _lib
ctx = _lib.Context(0)
rng = numpy.random.RandomState(42)
cases = {"flux4838": 1 + rng.normal(0, 5e-5, 4838), "flux300": 1 + rng.normal(0, 5e-5, 300), "flux64": 1 + rng.normal(0, 5e-5, 64),
         "flux20k": 1 + rng.normal(0, 2e-4, 19440), "ones": numpy.ones(3000), "halfs": numpy.full(2000, 0.5), "flux78k": 1 + rng.normal(0, 5e-5, 78544)}
for name, v in cases.items():
    want = numpy.concatenate([[0.0], numpy.cumsum(v)])
    for threads in (64, 512, 1024):
        got = ctx.debug_cumsum(v, threads=threads)
        ph = ctx.phase_cycles()
        keys = list(ph.keys())
        raw = [ph[k] for k in keys]
        print(name, threads, "exact", bool(numpy.array_equal(got, want)), "blocks", raw[10], "fails", raw[11], "specials", raw[23], "overflow", raw[18],
              "bad_lanes", raw[19], "wave_mismatch", raw[20], "t0", raw[21], "first_bad_thread", raw[22] if raw[22] < 2**63 else None, flush=True)
