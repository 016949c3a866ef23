"""Developer tool: prepare one BASELINE configuration (optionally every k-th period) and run its search a few
times -- the command rocprofv3 wraps for the kernel-trace and --pmc passes of tools/profile_config.sh.
    python tools/gpu_config_time.py tess_27d [stride] [repeats]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tls_amd import synthetic, _lib  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "tess_27d"
stride = int(sys.argv[2]) if len(sys.argv) > 2 else 1
repeats = int(sys.argv[3]) if len(sys.argv) > 3 else 3
ctx = _lib.Context(0)
t, f, kw = synthetic.config(name)
inp = synthetic.search_inputs(t, f, **kw)
sel = inp["periods"][::stride]
ctx.prepare(inp["t"], inp["y"], inp["dy"], sel, inp["table"], inp["params"])
for _ in range(min(repeats, 20)):   # (clocks up before the timed launches)
    ctx.execute()
ctx.synchronize()
ms = ctx.execute_timed(repeats)
print("%s: %d periods, n = %d, %.3f ms per launch" % (name, len(sel), len(inp["t"]), ms), flush=True)
