#!/bin/bash
# One GPU-box pass before a commit of measurements: parity suite (release + checked build), phase clocks,
# rocprofv3 passes of bench.py and of the Kepler-size sample, the bench line itself.
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r02}
OUT=$ROOT/gpurun_out/round_$TAG
mkdir -p "$OUT"
cd "$ROOT"
timeout 900 python -m pytest tests -m gpu -q -x > "$OUT/pytest.txt" 2>&1; grep -E "passed|failed" "$OUT/pytest.txt" | tail -2
TLS_AMD_DEBUG=1 TLS_AMD_LIB=$ROOT/tls_amd/libtls_amd_debug.so timeout 2400 python -m pytest tests -m gpu -q -x -s > "$OUT/pytest_debug.txt" 2>&1; grep -E "passed|failed" "$OUT/pytest_debug.txt" | tail -2
TLS_AMD_DEBUG=1 TLS_AMD_LIB=$ROOT/tls_amd/libtls_amd_clocks.so timeout 300 python tools/gpu_phases.py > "$OUT/phases.txt" 2>&1; cut -c1-160 "$OUT/phases.txt"
timeout 1200 bash tools/profile_gpu.sh $TAG > "$OUT/profile_gpu.log" 2>&1
timeout 900 bash tools/profile_kepler.sh $TAG > "$OUT/profile_kepler.log" 2>&1; tail -3 "$OUT/profile_kepler.log" | cut -c1-600
tail -1 gpurun_out/prof_$TAG/bench.json | cut -c1-1500
