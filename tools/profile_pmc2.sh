#!/bin/bash
# Second-level PMC passes for the search kernel (issue/stall mix); run via gpurun.
set -u
TAG=${1:-r02b}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BENCH="python $ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extras ${BENCH_ARGS:-}"
rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d "$OUT/p1" -o k -- $BENCH > "$OUT/p1.log" 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_SALU SQ_INSTS_SMEM --output-format csv -d "$OUT/p2" -o k -- $BENCH > "$OUT/p2.log" 2>&1
rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_LDS_LOAD SQ_INSTS_LDS_STORE SQ_INSTS_LDS_ATOMIC SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM --output-format csv -d "$OUT/p3" -o k -- $BENCH > "$OUT/p3.log" 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_SALU SQ_LEVEL_WAVES SQ_INSTS_BRANCH --output-format csv -d "$OUT/p4" -o k -- $BENCH > "$OUT/p4.log" 2>&1
python - <<PY
import csv, collections, glob
for d in ("p1","p2","p3","p4"):
    for f in glob.glob("$OUT/%s/*counter_collection.csv" % d):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "tls_search" in r["Kernel_Name"] or "tls_slim" in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in sorted(agg.items()):
            print("%-28s %.4g" % (k, sum(v)/len(v)))
PY
