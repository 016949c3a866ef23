#!/bin/bash
# A/B of library variants in ONE box: kernel time of the named configs under the shipped library and under each
# libtls_amd_<NAME>.so given (make -C tls_amd/csrc variant NAME=... DEFS=...), two interleaved passes.
#   tools/gpu_ab_many.sh "cfg1 cfg2 ..." NAME1 NAME2 ...
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
CFGS=$1; shift
cd "$ROOT"
for pass in 1 2; do
  for cfg in $CFGS; do
    python tools/gpu_ab_time.py $cfg 5
    for name in "$@"; do
      TLS_AMD_DEBUG=1 TLS_AMD_LIB=$ROOT/tls_amd/libtls_amd_$name.so python tools/gpu_ab_time.py $cfg 5
    done
  done
done
