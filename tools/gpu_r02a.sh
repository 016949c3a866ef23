#!/bin/bash
# Round-2 GPU check (run through gpurun): the -m gpu suite, per-phase clocks of two builds, the bench line.
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r02a}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$ROOT"
timeout 300 python -m pytest tests -m gpu -q -k "cumsum" > "$OUT/cumsum.txt" 2>&1
tail -3 "$OUT/cumsum.txt"
timeout 1200 python -m pytest tests -m gpu -q --durations=12 > "$OUT/pytest.txt" 2>&1
tail -25 "$OUT/pytest.txt"
timeout 300 python tools/gpu_phases.py > "$OUT/phases_new.txt" 2>&1
if [ -f tls_amd/libtls_amd_old.so ]; then
  TLS_AMD_LIB=$ROOT/tls_amd/libtls_amd_old.so timeout 300 python tools/gpu_phases.py > "$OUT/phases_old.txt" 2>&1
fi
cut -c1-400 "$OUT"/phases_*.txt
timeout 900 python bench.py > "$OUT/bench.json" 2> "$OUT/bench.err"
tail -c 3000 "$OUT/bench.json"; tail -5 "$OUT/bench.err"
