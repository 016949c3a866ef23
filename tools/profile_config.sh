#!/bin/bash
# Run on the GPU box (via gpurun): kernel duration and HBM traffic (separate --pmc passes) of one configuration.
#   tools/profile_config.sh <config> [stride] [tag]   -> gpurun_out/prof_<config>_<tag>.json
set -u
CFG=${1:-tess_27d}; STRIDE=${2:-1}; TAG=${3:-r02}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
CMD="python $ROOT/tools/gpu_config_time.py $CFG $STRIDE ${REPEATS:-40}"   # (many launches: the first few find the clocks down)
OUT=$ROOT/gpurun_out/prof_${CFG}_$TAG
mkdir -p "$OUT"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o k -- $CMD > "$OUT/trace.log" 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch" -o k -- $CMD > "$OUT/fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/write" -o k -- $CMD > "$OUT/write.log" 2>&1
python - <<PY > $ROOT/gpurun_out/prof_${CFG}_$TAG.json
import csv, glob, json, re
def mean(path, name):
    v = [float(r["Counter_Value"]) for f in glob.glob(path) for r in csv.DictReader(open(f))
         if "tls_search" in r["Kernel_Name"] and r["Counter_Name"] == name]
    return sum(v) / max(len(v), 1), len(v)
f, nf = mean("$OUT/fetch/*counter_collection.csv", "FETCH_SIZE")
w, nw = mean("$OUT/write/*counter_collection.csv", "WRITE_SIZE")
dur = [float(r["AverageNs"]) for f2 in glob.glob("$OUT/trace/*kernel_stats.csv") for r in csv.DictReader(open(f2)) if "tls_search" in r["Name"]]
m = re.search(r"(\d+) periods, n = (\d+)", open("$OUT/trace.log").read())
n_periods, n = int(m.group(1)), int(m.group(2))
algo = n_periods * (24 * n + 24)
b = (2 * f + w) * 1024   # gfx950: FETCH_SIZE counts 64 B per 128 B request (MI355X_MICROARCH.md)
print(json.dumps({"config": "$CFG", "stride": $STRIDE, "n_periods": n_periods, "n": n, "fetch_kib": f, "write_kib": w, "launches": [nf, nw],
      "kernel_ms": dur[0] * 1e-6 if dur else None, "bytes_per_launch": b, "algorithmic_bytes_per_launch": algo,
      "traffic_over_algorithmic": b / algo, "hbm_GBps": b / (dur[0] * 1e-9) / 1e9 if dur else None}))
PY
cat $ROOT/gpurun_out/prof_${CFG}_$TAG.json
