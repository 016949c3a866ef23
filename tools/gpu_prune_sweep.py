"""Developer tool: kernel time of the plain and the pruning variant over noise levels and pruning thresholds
(calibrates the host's choice, pruning_pays() in tls_amd/csrc/tls_amd.hip), every pruned result compared
bit for bit with the plain kernel's.
    python tools/gpu_prune_sweep.py [config[@ppm] ...]"""
import os
import sys

import numpy

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tls_amd import synthetic, _lib  # noqa: E402

ctx = _lib.Context(0)
cases = sys.argv[1:] or ["k2_90d@50", "k2_90d@100", "k2_90d@200", "k2_90d@500", "k2_90d@1000", "tutorial01@50"]
for case in cases:
    name, _, ppm = case.partition("@")
    sigma = float(ppm) * 1e-6 if ppm else None
    t, f, kw = synthetic.config(name, sigma=sigma)
    inp = synthetic.search_inputs(t, f, **kw)
    out, ref = [], None
    for mode, min_live in (("0", None), ("1", "0"), ("1", "256"), ("1", "1000"), ("1", "4000"), (None, None)):
        for k, v in (("TLS_PRUNE", mode), ("TLS_PRUNE_MIN_LIVE", min_live)):
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
        ctx.prepare(inp["t"], inp["y"], inp["dy"], inp["periods"], inp["table"], inp["params"])
        ctx.execute()
        got = ctx.fetch()
        if ref is None:
            ref = got
        same = all(numpy.array_equal(x, y) for x, y in zip(ref, got))
        out.append("%s/%s %.3f ms%s" % (mode, min_live, ctx.execute_timed(5), "" if same else " DIFFERENT"))
    print(case, " | ".join(out), flush=True)
