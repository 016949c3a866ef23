"""Developer sweep: slab-variant kernel time against the developer switch band_max (expected band hits above which a period starts
in exact mode), full grids and the first / last blocks of an 8-way shard."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy
from tls_amd import synthetic, _lib
ctx = _lib.Context(0)
for name, stride in (("tess_27d", 1), ("kepler_4yr", 16)):
    t, f, kw = synthetic.config(name)
    inp = synthetic.search_inputs(t, f, **kw)
    periods = inp["periods"][::stride]
    blocks = {"all": periods, "first_eighth": periods[: len(periods) // 8], "last_256": periods[-256:], "first_512": periods[:512]}
    for bm in (0.0, 0.1, 1.0, 10.0, 1e9):
        ctx.set_options(band_max=bm)
        out = []
        for label, per in blocks.items():
            ctx.prepare(inp["t"], inp["y"], inp["dy"], per, inp["table"], inp["params"])
            ctx.execute(); ctx.synchronize()
            ms = min(ctx.execute_timed(5 if len(per) * len(t) < 3e8 else 2) for _ in range(3))
            out.append("%s %.3f" % (label, ms))
        print(name, "band_max", bm, " ".join(out), flush=True)
