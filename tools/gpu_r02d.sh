#!/bin/bash
# parity suite + per-phase clocks (incl. the Kepler sample) + A/B of TLS_SORT3 + bench line
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r02d}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$ROOT"
timeout 900 python -m pytest tests -m gpu -q -x > "$OUT/pytest.txt" 2>&1; grep -E "passed|failed|Error|error" "$OUT/pytest.txt" | tail -5
timeout 300 python tools/gpu_phases.py k2_90d tess_27d kepler_4yr/64 > "$OUT/phases_new.txt" 2>&1
TLS_SORT3=0 timeout 300 python tools/gpu_phases.py tess_27d kepler_4yr/64 > "$OUT/phases_sort3off.txt" 2>&1
cat "$OUT"/phases_new.txt "$OUT"/phases_sort3off.txt | cut -c1-700
if [ "${2:-}" = "bench" ]; then
  timeout 900 python bench.py > "$OUT/bench.json" 2> "$OUT/bench.err"
  python - <<PY
import json
d=json.loads(open("$OUT/bench.json").read().strip().splitlines()[-1])
print("k2 ms_per_step", d["ms_per_step"], "kernel_ms", d["roofline"]["kernel_ms"], "one_shot", d["config"]["one_shot"], "power_ms", d["config"]["power_call_wall_ms_per_light_curve"])
for k in ("tess_27d","kepler_4yr","survey_1024"):
    print(k, {kk:vv for kk,vv in d[k].items() if kk in ("kernel_ms","trial_cells_per_s","curves_per_s","wall_s","argmin_period_index","error")}, d[k].get("roofline",{}).get("frac"))
print("fp64", d["roofline"]["fp64"])
PY
fi
