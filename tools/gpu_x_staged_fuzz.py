import os, sys
import numpy
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from tls_amd import synthetic, _lib, transit_model
ctx = _lib.Context(0)
rng = numpy.random.RandomState(12)
bad = 0
for case in range(10):
    n = int(rng.choice([12000, 15000, 20000, 30002, 44000]))
    span = float(rng.choice([20.0, 40.0, 80.0]))
    t = numpy.sort(rng.uniform(0.5, 0.5 + span, n)) if case % 3 == 0 else numpy.linspace(0.5, 0.5 + span, n)
    per = float(rng.uniform(2.0, span / 5))
    y = transit_model.light_curve(t, 1.3, per, float(rng.uniform(0.01, 0.06)), 12, 89.8, 0, 90, [0.4, 0.3], "quadratic") + rng.normal(0, float(rng.choice([1e-4, 5e-4, 2e-3])), n)
    dy = rng.uniform(0.7, 1.4, n) * 3e-4 if case % 2 else None
    inp = synthetic.search_inputs(t, y, dy, period_min=1.0, period_max=span / 3, oversampling_factor=1)
    sel = inp["periods"][::max(1, len(inp["periods"]) // 2600)]
    out = {}
    for flag in ("0", "1"):
        ctx.set_options(x_staged=flag, split=0)
        out[flag] = ctx.search(inp["t"], inp["y"], inp["dy"], sel, inp["table"], inp["params"], count_work=True)
    info = ctx.plan_info()
    same_rows = numpy.array_equal(out["0"][1], out["1"][1])
    rel = float(numpy.max(numpy.abs(out["0"][0] - out["1"][0]) / out["0"][0]))
    cells = out["0"][3]["evaluated_cells"] == out["1"][3]["evaluated_cells"]
    ok = same_rows and rel <= 1e-12 and cells
    bad += 0 if ok else 1
    print("case", case, "n", len(inp["t"]), "periods", len(sel), "resident", info["resident"], "weights", dy is not None, "rows", same_rows, "rel %.2e" % rel, "cells", cells, flush=True)
print("BAD", bad)
