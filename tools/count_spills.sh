#!/bin/bash
# Developer tool: instructions / scratch (spill) instructions / FLAT instructions per device function of the
# search kernels' assembly.  usage: tools/count_spills.sh [extra hipcc flags]; leaves the assembly in /tmp/tls.s
SRC="$(cd "$(dirname "$0")/../tls_amd/csrc" && pwd)"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-fast-math -S --cuda-device-only "$@" -o /tmp/tls.s "$SRC/tls_amd.hip" 2>&1 | grep -E "error" -A4
awk '/^_ZN6tlsdev.*:/ {name=$1} /scratch_load|scratch_store/ {n[name]++} /^\t(v_|s_|ds_|global_|scratch_|buffer_|flat_)/ {t[name]++} /^\tflat_/ {f[name]++}
     END {for (k in t) printf "%7d instrs %5d scratch %4d flat  %s\n", t[k], n[k], f[k], substr(k, 1, 90)}' /tmp/tls.s | sort -k7
