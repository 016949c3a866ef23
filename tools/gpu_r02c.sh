#!/bin/bash
# quick A/B: cumsum diagnostics, the cumsum test, per-phase clocks of the current and the "old" build
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r02c}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$ROOT"
if [ -f tls_amd/libtls_amd_diag.so ]; then
  TLS_AMD_LIB=$ROOT/tls_amd/libtls_amd_diag.so timeout 120 python tools/gpu_cumsum_diag.py > "$OUT/diag.txt" 2>&1; cat "$OUT/diag.txt" | cut -c1-200
fi
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x > "$OUT/pytest.txt" 2>&1; tail -4 "$OUT/pytest.txt"
timeout 300 python tools/gpu_phases.py > "$OUT/phases_new.txt" 2>&1
if [ -f tls_amd/libtls_amd_old.so ]; then
  TLS_AMD_LIB=$ROOT/tls_amd/libtls_amd_old.so timeout 300 python tools/gpu_phases.py > "$OUT/phases_old.txt" 2>&1
fi
python - <<PY
import ast,re,glob
for f in sorted(glob.glob("$OUT/phases_*.txt")):
    print(f.split('/')[-1])
    for line in open(f):
        m=re.match(r"(\S+) (\S+) ([\d.]+) ms cells/s (\S+) (\{.*\}) cycles/period/wg (\d+) cumsum blocks (\d+) fallbacks (\d+)", line)
        if not m: print(line[:200]); continue
        d=ast.literal_eval(m.group(5))
        keep={k:v for k,v in d.items() if float(v[:-1])>=0.4}
        print(m.group(1),m.group(2),m.group(3),"ms cyc",m.group(6),"fb",m.group(8)," ".join("%s=%s"%(k,v[:-1]) for k,v in keep.items()))
PY
