#!/bin/bash
# Same-box A/B of the four-slot kernel (TLS_SLIM=1) against the classic LDS-resident one (TLS_SLIM=0), interleaved.
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd "$ROOT"
for pass in 1 2 3; do
  for cfg in ${1:-k2_90d}; do
    TLS_SLIM=0 python tools/gpu_ab_time.py $cfg 5 | sed 's/$/  [classic]/'
    TLS_SLIM=1 python tools/gpu_ab_time.py $cfg 5 | sed 's/$/  [slim]/'
  done
done
