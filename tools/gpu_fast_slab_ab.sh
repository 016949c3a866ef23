#!/bin/bash
# Same-box A/B of the slab variant's prefix-sum modes: exact (default) against TLS_FAST_SLAB=1, for the shipped
# library and for libtls_amd_<NAME>.so (make variant NAME=...).   tools/gpu_fast_slab_ab.sh [NAME] [configs...]
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
NAME=${1:-}; shift
CFGS=${*:-tess_27d kepler_4yr/64}
cd "$ROOT"
for rep in 1 2; do
  for cfg in $CFGS; do
    for fast in 0 1; do
      TLS_FAST_SLAB=$fast timeout 300 python tools/gpu_ab_time.py $cfg 5 | sed "s/^/fast=$fast /"
      if [ -n "$NAME" ]; then
        TLS_FAST_SLAB=$fast TLS_AMD_DEBUG=1 TLS_AMD_LIB=$ROOT/tls_amd/libtls_amd_$NAME.so timeout 300 python tools/gpu_ab_time.py $cfg 5 | sed "s/^/fast=$fast /"
      fi
    done
  done
done
for fast in 0 1; do
  echo "== phases, TLS_FAST_SLAB=$fast"
  TLS_FAST_SLAB=$fast TLS_AMD_DEBUG=1 TLS_AMD_LIB=$ROOT/tls_amd/libtls_amd_clocks.so timeout 300 python tools/gpu_phases.py $CFGS 2>&1 | cut -c1-900
done
