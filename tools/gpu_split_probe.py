"""Developer tool (GPU box): short launches of a slab series through the one-workgroup kernel and the two-role kernel
(fast mode, row parts 1..3): kernel time per block and bit equality with the full-grid search.
    python tools/gpu_split_probe.py [config] [n_blocks]"""
import os
import sys
import time

import numpy

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tls_amd import synthetic, _lib, shard  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "tess_27d"
n_blocks = int(sys.argv[2]) if len(sys.argv) > 2 else 8
stride = int(sys.argv[3]) if len(sys.argv) > 3 else 1
ctx = _lib.Context(0)
t, f, kw = synthetic.config(name)
inp = synthetic.search_inputs(t, f, **kw)
periods = inp["periods"][::stride]
ctx.set_options(split=0)
whole = ctx.search(inp["t"], inp["y"], inp["dy"], periods, inp["table"], inp["params"])
for _ in range(30):   # (clocks up)
    ctx.execute()
ctx.synchronize()
whole_ms = ctx.execute_timed(10)
print(name, len(periods), "periods, whole grid %.3f ms (%s)" % (whole_ms, ctx.last_kernel()), flush=True)
job = shard.ShardedSearch(0, n_blocks, layout="blocks")
job.plan(inp["t"], periods, inp["table"], inp["params"], y=inp["y"], options=ctx.get_options())
bounds = job.bounds
if os.environ.get("PARTITION") == "sum":
    bounds = shard.partition_by_cost(job.times, n_blocks)
cyclic = os.environ.get("PARTITION") == "cyclic"
print("blocks", "cyclic" if cyclic else list(numpy.diff(bounds)))
for label, sw in (("one-wg", dict(split=0)), ("split", dict(split=1)), ("auto", dict(split=None))):
    ctx.set_options(**sw)
    ms, bad = [], 0
    for r in range(n_blocks):
        lo, hi = int(bounds[r]), int(bounds[r + 1])
        sel = slice(r, None, n_blocks) if cyclic else slice(lo, hi)
        got = ctx.search(inp["t"], inp["y"], inp["dy"], periods[sel], inp["table"], inp["params"])
        for a, b in zip(got[:3], whole[:3]):
            bad += int(numpy.sum(a != b[sel]))
        ctx.execute()
        ctx.synchronize()
        ms.append(ctx.execute_timed(5))
    print("%-9s %-11s max %.3f mean %.3f speedup %.2f mismatches %d  %s" % (label, ctx.last_kernel(), max(ms), sum(ms) / len(ms), whole_ms / max(ms), bad,
                                                                    " ".join("%.3f" % m for m in ms)), flush=True)
