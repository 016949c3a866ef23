"""Developer tool: kernel time of one configuration, interleaved repetitions (for A/B runs of library variants).
    TLS_AMD_DEBUG=1 TLS_AMD_LIB=... python tools/gpu_ab_time.py [config[@ppm][/period stride]] [reps]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tls_amd import synthetic, _lib  # noqa: E402

case = sys.argv[1] if len(sys.argv) > 1 else "k2_90d"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
case_name, _, stride = case.partition("/")
name, _, ppm = case_name.partition("@")
t, f, kw = synthetic.config(name, sigma=float(ppm) * 1e-6 if ppm else None)
inp = synthetic.search_inputs(t, f, **kw)
ctx = _lib.Context(0)
periods = inp["periods"][::int(stride)] if stride else inp["periods"]
ctx.prepare(inp["t"], inp["y"], inp["dy"], periods, inp["table"], inp["params"])
ctx.execute(); ctx.synchronize()
times = [ctx.execute_timed(20 if len(periods) * len(inp["t"]) < 3e8 else 3) for _ in range(reps)]
print(case, os.environ.get("TLS_AMD_LIB", "default").split("/")[-1], "min %.4f median %.4f ms" % (min(times), sorted(times)[len(times) // 2]), flush=True)
