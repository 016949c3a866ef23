import sys, time, warnings
sys.path.insert(0, '/root/repo')
import numpy, tls_amd
from tls_amd import synthetic, _lib, search as ts
t, f = synthetic.light_curve(30.0, 48, 2e-4, per=4.321, rp=0.05, a=12)
kw = dict(period_min=1.0, period_max=9.0, oversampling_factor=2, show_progress_bar=False, verbose=False)
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    plain = tls_amd.transitleastsquares(t, f, verbose=False).power(**kw)
    shard = tls_amd.transitleastsquares(t, f, verbose=False).power(devices=[0, 0], **kw)
for k in plain.keys():
    a, b = numpy.asarray(plain[k], dtype=float), numpy.asarray(shard[k], dtype=float)
    if a.shape != b.shape or not numpy.array_equal(a, b, equal_nan=True):
        print("DIFF", k, a.ravel()[:3], b.ravel()[:3], numpy.max(numpy.abs(a - b)) if a.shape == b.shape else "shape")
# timing breakdown of the fused call
t, f, kw = synthetic.config("k2_90d")
ctx = ts.default_context(None)
m = tls_amd.transitleastsquares(t, f, verbose=False)
for _ in range(5): m.power(verbose=False, show_progress_bar=False, **kw)
inp = synthetic.search_inputs(t, f, **kw)
best = 1e9
for _ in range(20):
    t0 = time.perf_counter(); out = ts.fused_power(inp["t"], inp["y"], inp["dy"], inp["periods"], inp["table"], inp["params"], 3); best = min(best, time.perf_counter() - t0)
print("fused_power best %.3f ms" % (best * 1e3))
best = 1e9
for _ in range(20):
    t0 = time.perf_counter(); ctx.search(inp["t"], inp["y"], inp["dy"], inp["periods"], inp["table"], inp["params"]); best = min(best, time.perf_counter() - t0)
print("ctx.search best %.3f ms" % (best * 1e3))
best = 1e9
for _ in range(20):
    t0 = time.perf_counter(); m.power(verbose=False, show_progress_bar=False, **kw); best = min(best, time.perf_counter() - t0)
print("power() best %.3f ms" % (best * 1e3))
print("group ms", ctx.batch_group_ms())
