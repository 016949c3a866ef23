"""Developer probe: which periods of a sharded search (two contexts on GPU 0) differ from the one-context search, and
what the oracle says about them."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy
from tls_amd import synthetic, _lib, search as tsearch
import oracle

name = sys.argv[1] if len(sys.argv) > 1 else "k2_90d"
t, f, kw = synthetic.config(name)
inp = synthetic.search_inputs(t, f, **kw)
one = _lib.Context(0)
want = one.search(inp["t"], inp["y"], inp["dy"], inp["periods"], inp["table"], inp["params"])
again = one.search(inp["t"], inp["y"], inp["dy"], inp["periods"], inp["table"], inp["params"])
print("repeat equal", numpy.array_equal(want[0], again[0]))
group = tsearch.DeviceGroup([0, 0])
got = group.search(inp["t"], inp["y"], inp["dy"], inp["periods"], inp["table"], inp["params"])
bad = numpy.nonzero(got[0] != want[0])[0]
print("blocks", group.last_blocks, "differing", bad)
p = inp["params"]
for sw in ({}, {"exact_prefix": 1}):
    one.set_options(exact_prefix=None); one.set_options(**sw)
    for lo, hi in ((0, len(inp["periods"])), (0, int(group.last_blocks[1])), (int(group.last_blocks[1]), len(inp["periods"]))):
        r = one.search(inp["t"], inp["y"], inp["dy"], inp["periods"][lo:hi], inp["table"], inp["params"])
        for b in bad:
            if lo <= b < hi:
                print(sw, (lo, hi), "period", b, "chi2 %.9f row %d depth %.12f" % (r[0][b - lo], r[1][b - lo], r[2][b - lo]))
if len(bad):
    o = oracle.search(inp["t"], inp["y"], inp["dy"], inp["periods"][bad], inp["table"], p["transit_depth_min"], p["R_star_min"],
                      p["R_star_max"], p["M_star_min"], p["M_star_max"], p["T0_fit_margin"])
    for k, b in enumerate(bad):
        print("oracle period", b, "chi2 %.9f row %d depth %.12f" % (o[0][k], o[1][k], o[2][k]), "one", want[0][b], "sharded", got[0][b])
