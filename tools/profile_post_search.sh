#!/bin/bash
# Run on the GPU box (via gpurun): rocprofv3 kernel stats and one PMC pass of the post-search kernels (final T0 fit
# stats.py:135-204, spectra + running median stats.py:105-132) -- a loop of the drop-in power() call and a
# tls_power_batch survey group.   tools/profile_post_search.sh <tag>  -> gpurun_out/prof_post_<tag>/
set -u
TAG=${1:-r05}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/prof_post_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/power" -o k -- python $ROOT/tools/gpu_power_profile.py > "$OUT/power.log" 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/batch" -o k -- python $ROOT/tools/gpu_power_batch_time.py 256 > "$OUT/batch.log" 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_SALU --output-format csv -d "$OUT/pmc" -o k -- python $ROOT/tools/gpu_power_profile.py > "$OUT/pmc.log" 2>&1
python - <<PY
import csv, glob, collections
for d in ("power", "batch"):
    for f in glob.glob("$OUT/%s/*kernel_stats.csv" % d):
        print("==", d)
        for r in csv.DictReader(open(f)):
            print("%-90s calls %6s avg_us %10.1f total_ms %9.2f pct %5s" % (r["Name"][:90], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6, r["Percentage"]))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/pmc/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, c in agg.items():
    print("PMC", k, " ".join("%s=%.4g" % (n, sum(v) / len(v)) for n, v in sorted(c.items())))
tail = open("$OUT/power.log").read().strip().splitlines()[-3:] + open("$OUT/batch.log").read().strip().splitlines()[-2:]
print("\n".join(tail))
PY
