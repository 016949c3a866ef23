"""Developer tool (GPU box): fit the shard cost model of tls_amd/shard.py to measured per-period shader cycles.

    python tools/gpu_cost_model.py [config ...]     (default: k2_90d tess_27d kepler_4yr/16)

For every configuration: tls_debug_period_cycles gives the cycles the workgroup of each period spent; the model is
    cycles(p) = a * N + b * cells(p) + c * taps(p)
(N points: fold, sort, prefix sum; cells: depth predicate; taps: sliding chi^2), least squares over the periods.
Prints the coefficients per kernel variant (LDS-resident / HBM slab), the fit quality, and the time imbalance
(max/mean of summed measured cycles) of G = 2, 4, 8 contiguous blocks placed by (i) cells only, (ii) the model."""
import json
import os
import sys

import numpy

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tls_amd import synthetic, _lib, shard  # noqa: E402

ctx = _lib.Context(0)
cases = sys.argv[1:] or ["k2_90d", "tess_27d", "kepler_4yr/16"]
out = {}
for case in cases:
    name, _, stride = case.partition("/")
    name, _, ppm = name.partition("@")
    t, f, kw = synthetic.config(name, sigma=float(ppm) * 1e-6 if ppm else None)
    inp = synthetic.search_inputs(t, f, **kw)
    periods = inp["periods"][::int(stride)] if stride else inp["periods"]
    ctx.prepare(inp["t"], inp["y"], inp["dy"], periods, inp["table"], inp["params"])
    ctx.execute()
    ctx.synchronize()
    cyc = numpy.median([ctx.period_cycles().astype(float) for _ in range(3)], axis=0)
    cells, taps, model_time = _lib.period_costs(inp["t"], periods, inp["table"], inp["params"], float(numpy.std(inp["y"])))
    n = len(inp["t"])
    A = numpy.stack([numpy.full(len(periods), float(n)), cells.astype(float), taps], axis=1)
    coef, *_ = numpy.linalg.lstsq(A, cyc, rcond=None)
    fit = A @ coef
    resid = (cyc - fit) / cyc
    rec = {"points": n, "periods": len(periods), "resident": ctx.plan_info()["resident"], "kernel": ctx.last_kernel(),
           "a_per_point": coef[0], "b_per_cell": coef[1], "c_per_tap": coef[2],
           "fixed_share": coef[0] * n * len(periods) / cyc.sum(), "cells_share": float((coef[1] * cells).sum() / cyc.sum()),
           "taps_share": float((coef[2] * taps).sum() / cyc.sum()),
           "rel_residual_rms": float(numpy.sqrt(numpy.mean(resid ** 2))), "rel_residual_max": float(numpy.abs(resid).max())}
    for G in (2, 4, 8):
        for label, cost in (("cells", cells.astype(float)), ("model", model_time)):
            b = shard.partition_by_cost(cost, G)
            blocks = numpy.array([cyc[b[r]:b[r + 1]].sum() for r in range(G)])
            rec["imbalance_G%d_%s" % (G, label)] = float(blocks.max() / blocks.mean())
    out[case] = rec
    print(case, json.dumps(rec), flush=True)
    os.makedirs("gpurun_out", exist_ok=True)
    numpy.savez_compressed("gpurun_out/cost_model_data_%s%s.npz" % (case.replace("/", "_"), ("_slim" + os.environ["TLS_SLIM"]) if os.environ.get("TLS_SLIM") else ""), cycles=cyc, cells=cells, taps=taps,
                           periods=periods, n=n, resident=rec["resident"], sigma=float(numpy.std(inp["y"])))
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/cost_model_fit.json", "w"), indent=1)
