#!/bin/bash
# Run on the GPU box (via gpurun): HBM traffic of the Kepler-size configuration (BASELINE config 3),
# a spread sample of its periods (every 64th): separate --pmc passes, kernel trace for the duration -> gpurun_out/prof_kepler_<tag>.json;
# and the rocprofv3 kernel-stats record of the FULL grid (182 388 periods, three launches) -> gpurun_out/prof_kepler_<tag>_full/.
set -u
TAG=${1:-r02}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
CMD="python $ROOT/tools/gpu_kepler_time.py 64"
echo "[" > $ROOT/gpurun_out/prof_kepler_$TAG.json
SEP=""
for VAR in default; do
  OUT=$ROOT/gpurun_out/prof_kepler_${TAG}_$VAR
  mkdir -p "$OUT"
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o k -- $CMD > "$OUT/trace.log" 2>&1
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch" -o k -- $CMD > "$OUT/fetch.log" 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/write" -o k -- $CMD > "$OUT/write.log" 2>&1
  python - <<PY >> $ROOT/gpurun_out/prof_kepler_$TAG.json
import csv, glob, json
def mean(path, name):
    v = [float(r["Counter_Value"]) for f in glob.glob(path) for r in csv.DictReader(open(f))
         if "tls_search" in r["Kernel_Name"] and r["Counter_Name"] == name]
    return sum(v) / max(len(v), 1), len(v)
f, nf = mean("$OUT/fetch/*counter_collection.csv", "FETCH_SIZE")
w, nw = mean("$OUT/write/*counter_collection.csv", "WRITE_SIZE")
dur = [float(r["AverageNs"]) for f2 in glob.glob("$OUT/trace/*kernel_stats.csv") for r in csv.DictReader(open(f2)) if "tls_search" in r["Name"]]
n_periods = 2850
algo = n_periods * (24 * 70128 + 24)
b = (2 * f + w) * 1024   # gfx950: FETCH_SIZE counts 64 B per 128 B request (MI355X_MICROARCH.md)
print("$SEP" + json.dumps({"name": "$VAR", "n_periods": n_periods, "fetch_kib": f, "write_kib": w, "launches": [nf, nw],
      "kernel_ms": dur[0] * 1e-6 if dur else None, "bytes_per_launch": b, "bytes_per_period": b / n_periods,
      "algorithmic_bytes_per_launch": algo, "traffic_over_algorithmic": b / algo,
      "hbm_GBps": b / (dur[0] * 1e-9) / 1e9 if dur else None}))
PY
  SEP=","
done
echo "]" >> $ROOT/gpurun_out/prof_kepler_$TAG.json
cat $ROOT/gpurun_out/prof_kepler_$TAG.json
FULL=$ROOT/gpurun_out/prof_kepler_${TAG}_full
mkdir -p "$FULL"
rocprofv3 --kernel-trace --stats --output-format csv -d "$FULL/trace" -o k -- python $ROOT/tools/gpu_config_time.py kepler_4yr 1 2 > "$FULL/trace.log" 2>&1
tail -1 "$FULL/trace.log"; grep -h "tls_search" "$FULL"/trace/*kernel_stats.csv | cut -c1-200
