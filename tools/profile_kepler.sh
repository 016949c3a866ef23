#!/bin/bash
# Run on the GPU box (via gpurun): HBM traffic of the Kepler-size configuration (BASELINE config 3),
# a spread sample of its periods.  Separate --pmc passes, kernel trace for the duration.
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/prof_kepler
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
CMD="python $ROOT/tools/gpu_kepler_time.py 64"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o k -- $CMD > "$OUT/trace.log" 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch" -o k -- $CMD > "$OUT/fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/write" -o k -- $CMD > "$OUT/write.log" 2>&1
python - <<PY
import csv, glob
def mean(path, name):
    v = [float(r["Counter_Value"]) for f in glob.glob(path) for r in csv.DictReader(open(f))
         if "tls_search" in r["Kernel_Name"] and r["Counter_Name"] == name]
    return sum(v) / max(len(v), 1), len(v)
f, nf = mean("$OUT/fetch/*counter_collection.csv", "FETCH_SIZE")
w, nw = mean("$OUT/write/*counter_collection.csv", "WRITE_SIZE")
dur = [float(r["AverageNs"]) for f2 in glob.glob("$OUT/trace/*kernel_stats.csv") for r in csv.DictReader(open(f2)) if "tls_search" in r["Name"]]
print("launches", nf, nw, "FETCH_SIZE KiB", f, "WRITE_SIZE KiB", w, "kernel avg ns", dur)
if dur:
    bytes_ = (2 * f + w) * 1024   # gfx950: FETCH_SIZE counts 64 B per 128 B request (MI355X_MICROARCH.md)
    print("HBM bytes per launch %.3e -> %.1f GB/s" % (bytes_, bytes_ / (dur[0] * 1e-9) / 1e9))
PY
