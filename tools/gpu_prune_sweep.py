"""Developer tool: kernel time of the plain and the pruning variant over noise levels (calibrates
the host's choice, pruning_pays() in tls_amd/csrc/tls_amd.hip)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tls_amd import synthetic, _lib  # noqa: E402

ctx = _lib.Context(0)
for name in ("k2_90d", "tess_27d"):
    for sigma in (50e-6, 100e-6, 150e-6, 200e-6, 300e-6, 500e-6, 1000e-6):
        t, f, kw = synthetic.config(name, sigma=sigma)
        inp = synthetic.search_inputs(t, f, **kw)
        out = []
        for mode in ("0", "1", None):
            if mode is None:
                os.environ.pop("TLS_PRUNE", None)
            else:
                os.environ["TLS_PRUNE"] = mode
            ctx.prepare(inp["t"], inp["y"], inp["dy"], inp["periods"], inp["table"], inp["params"])
            ctx.execute(); ctx.synchronize()
            out.append(ctx.execute_timed(5))
        print("%s sigma %4.0f ppm: plain %.3f ms, pruning %.3f ms, auto %.3f ms" % (name, 1e6 * sigma, *out), flush=True)
