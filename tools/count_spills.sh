#!/bin/bash
# scratch (spill) instructions per kernel of the device assembly: tools/count_spills.sh [extra hipcc flags]
cd "$(dirname "$0")/../tls_amd/csrc" && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-fast-math -S --cuda-device-only "$@" -o /tmp/tls.s tls_amd.hip 2>&1 | grep -E "error"
awk '/^_ZN6tlsdev.*:/ {name=$1} /scratch_load|scratch_store/ {n[name]++} /^\s+v_|^\s+s_|^\s+ds_|^\s+global_|^\s+scratch_|^\s+buffer_|^\s+flat_/ {t[name]++} END {for (k in t) printf "%7d instrs %5d scratch  %s\n", t[k], n[k], k}' /tmp/tls.s | sort -k5 | grep search_kernel
