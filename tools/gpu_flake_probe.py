import sys, numpy
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
from conftest import load_search_golden, SEARCH_GOLDENS
from tls_amd import _lib
ctx = _lib.Context(0)
gold = {n: load_search_golden(n) for n in SEARCH_GOLDENS}
bad = 0
for it in range(300):
    for name in SEARCH_GOLDENS:
        g, table, params = gold[name]
        a = ctx.search(g["t"], g["y"], g["dy"], g["periods"], table, params, count_work=True)
        b = ctx.search(g["t"], g["y"], g["dy"], g["periods"], table, params)
        for tag, r in (("count", a), ("plain", b)):
            fin = numpy.isfinite(g["chi2"])
            rel = numpy.abs(r[0][fin] - g["chi2"][fin]) / numpy.abs(g["chi2"][fin])
            if rel.max() > 1e-6 or not numpy.array_equal(r[1], g["row"]):
                bad += 1
                print("MISMATCH", it, name, tag, ctx.last_kernel(), rel.max(), int(numpy.argmax(rel)), flush=True)
print("done, mismatches:", bad)
