"""Developer tool: per-phase cycle breakdown of the search kernel on a GPU box.  Needs the instrumented library:
    make -C tls_amd/csrc clocks
    TLS_AMD_DEBUG=1 TLS_AMD_LIB=$PWD/tls_amd/libtls_amd_clocks.so python tools/gpu_phases.py [config ...]
(default: k2_90d k2_90d@500 tutorial01 tess_27d kepler_4yr/64; the shipped library carries no phase clocks)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tls_amd import synthetic, _lib  # noqa: E402

ctx = _lib.Context(0)
cases = sys.argv[1:] or ["k2_90d", "k2_90d@500", "tutorial01", "tess_27d", "kepler_4yr/64"]
for case in cases:
    name, _, stride = case.partition("/")
    name, _, ppm = name.partition("@")
    sigma = float(ppm) * 1e-6 if ppm else None
    t, f, kw = synthetic.config(name, sigma=sigma)
    inp = synthetic.search_inputs(t, f, **kw)
    periods = inp["periods"][::int(stride)] if stride else inp["periods"]
    ctx.prepare(inp["t"], inp["y"], inp["dy"], periods, inp["table"], inp["params"])
    ctx.execute()
    ctx.synchronize()
    ms = ctx.execute_timed(3)
    ctx.execute(phase_clock=True)
    ph = ctx.phase_cycles()
    blocks, fails = ph.pop("cumsum_blocks"), ph.pop("cumsum_fallbacks")
    stats = {k: ph.pop(k) for k in list(ph) if k.startswith("stat_")}
    tot = sum(ph.values())
    if tot == 0:
        sys.exit("no phase clocks in this library: build `make -C tls_amd/csrc clocks` and select it with "
                 "TLS_AMD_DEBUG=1 TLS_AMD_LIB=.../libtls_amd_clocks.so")
    info = ctx.plan_info()
    print(case, "%.3f ms" % ms, "cells/s %.3e" % (info["grid_cells"] / ms * 1e3),
          "cyc/period/wg %.0f" % (tot / len(periods)), "cumsum blocks %d fallbacks %d |" % (blocks, fails),
          " ".join("%s=%.1f" % (k, 100.0 * v / tot) for k, v in ph.items() if v >= 0.004 * tot),
          "| per period:", " ".join("%s=%.1f" % (k[5:], v / len(periods)) for k, v in stats.items() if v), flush=True)
