#!/bin/bash
# Two-kernel slab path (fold kernel + search kernel over (period, tile) items) against the one-kernel path (TLS_SPLIT=0)
# on one box: slab parity tests, then kernel time and phase clocks of the slab configurations both ways.
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-split}; shift || true
CFGS=${*:-tess_27d kepler_4yr/64}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$ROOT"
timeout 1200 python -m pytest tests -m gpu -q -x -k "tess or kepler or slab or quarter or large_series or tiled or sort_order or randomised or commensurate or search_batch" > "$OUT/pytest.txt" 2>&1; tail -5 "$OUT/pytest.txt"
for rep in 1 2; do
  for c in $CFGS; do
    echo "== split"; timeout 300 python tools/gpu_ab_time.py $c 3 2>&1 | tail -1
    echo "== one kernel"; TLS_SPLIT=0 timeout 300 python tools/gpu_ab_time.py $c 3 2>&1 | tail -1
  done
done | tee "$OUT/ab.txt"
echo "== phases split"; TLS_AMD_DEBUG=1 TLS_AMD_LIB=$ROOT/tls_amd/libtls_amd_clocks.so timeout 300 python tools/gpu_phases.py $CFGS 2>&1 | cut -c1-900 | tee "$OUT/phases_split.txt"
echo "== phases one kernel"; TLS_SPLIT=0 TLS_AMD_DEBUG=1 TLS_AMD_LIB=$ROOT/tls_amd/libtls_amd_clocks.so timeout 300 python tools/gpu_phases.py $CFGS 2>&1 | cut -c1-900 | tee "$OUT/phases_one.txt"
