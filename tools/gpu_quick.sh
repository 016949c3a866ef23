#!/bin/bash
# parity suite + per-phase clocks of the named configs (default: the two slab configs)
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-quick}; shift || true
CFGS=${*:-tess_27d kepler_4yr/64}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$ROOT"
timeout 900 python -m pytest tests -m gpu -q -x > "$OUT/pytest.txt" 2>&1; grep -E "passed|failed|Error|error" "$OUT/pytest.txt" | tail -5
TLS_AMD_DEBUG=1 TLS_AMD_LIB=$ROOT/tls_amd/libtls_amd_clocks.so timeout 300 python tools/gpu_phases.py $CFGS > "$OUT/phases.txt" 2>&1
cut -c1-900 "$OUT/phases.txt"
