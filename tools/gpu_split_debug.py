"""Developer tool: the two-kernel slab path on a small sample, against the one-kernel path."""
import os
import sys
import numpy
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tls_amd import synthetic, _lib  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "tess_27d"
stride = int(sys.argv[2]) if len(sys.argv) > 2 else 100
t, f, kw = synthetic.config(name)
inp = synthetic.search_inputs(t, f, **kw)
periods = inp["periods"][::stride]
ctx = _lib.Context(0)
print("prepare", len(periods), "periods", flush=True)
ctx.prepare(inp["t"], inp["y"], inp["dy"], periods, inp["table"], inp["params"])
print("execute", os.environ.get("TLS_SPLIT"), os.environ.get("TLS_SPLIT_ONLY"), flush=True)
ctx.execute(); ctx.synchronize()
print("done", flush=True)
chi2, row, depth = ctx.fetch()[:3]
numpy.save(sys.argv[3], numpy.stack([chi2, row.astype(float), depth])) if len(sys.argv) > 3 else None
print("chi2 sum %.12f min %.12f argmin %d" % (chi2.sum(), chi2.min(), chi2.argmin()), flush=True)
