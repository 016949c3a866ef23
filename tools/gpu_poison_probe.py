"""Developer probe: searches before and after the LDS of every CU is filled with NaN words must agree bit for bit."""
import sys, os, numpy
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from conftest import load_search_golden, SEARCH_GOLDENS
from tls_amd import _lib, synthetic
ctx = _lib.Context(0)
cases = []
for name in SEARCH_GOLDENS:
    g, table, params = load_search_golden(name)
    cases.append(("golden " + name, (g["t"], g["y"], g["dy"], g["periods"], table, params)))
for name, sigma, stride, weights in (("k2_90d", None, 20, False), ("k2_90d", 200e-6, 20, False), ("k2_90d", 500e-6, 20, False),
                                     ("k2_90d", None, 20, True), ("tutorial01", None, 30, False), ("tess_27d", None, 60, False),
                                     ("tess_27d", None, 60, True), ("kepler_4yr", None, 4000, False)):
    t, f, kw = synthetic.config(name, sigma=sigma)
    dy = numpy.random.RandomState(5).uniform(0.7, 1.5, len(f)) * synthetic.CONFIGS[name][2] if weights else None
    inp = synthetic.search_inputs(t, f, dy, **kw)
    cases.append(("%s sigma=%s weights=%s" % (name, sigma, weights), (inp["t"], inp["y"], inp["dy"], inp["periods"][::stride], inp["table"], inp["params"])))
bad = 0
for label, args in cases:
    for word in (0x7ff80000, 0xfff00000, 0xffffffff):
        for count in (False, True):
            a = ctx.search(*args, count_work=count)
            ctx.poison_lds(word)
            b = ctx.search(*args, count_work=count)
            same = all(numpy.array_equal(x, y) for x, y in zip(a[:3], b[:3]))
            if not same:
                bad += 1
                d = numpy.nonzero(a[0] != b[0])[0]
                print("DIFFERS", label, hex(word), "count" if count else "plain", ctx.last_kernel(), len(d), "periods", d[:5], flush=True)
print("cases", len(cases), "mismatches", bad)
