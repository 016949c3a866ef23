#!/bin/bash
# A/B of two builds in one box: phases of the named configs under the instrumented library (make clocks) and under
# libtls_amd_<NAME>.so (build it with DEFS="... -DTLS_PHASE_CLOCKS=1")
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
NAME=$1; shift
CFGS=${*:-tess_27d kepler_4yr/64}
cd "$ROOT"; mkdir -p gpurun_out/ab_$NAME
for rep in 1 2; do
  echo "== default"; TLS_AMD_DEBUG=1 TLS_AMD_LIB=$ROOT/tls_amd/libtls_amd_clocks.so timeout 300 python tools/gpu_phases.py $CFGS 2>&1 | cut -c1-700
  echo "== $NAME"; TLS_AMD_DEBUG=1 TLS_AMD_LIB=$ROOT/tls_amd/libtls_amd_$NAME.so timeout 300 python tools/gpu_phases.py $CFGS 2>&1 | cut -c1-700
done | tee gpurun_out/ab_$NAME/phases.txt
