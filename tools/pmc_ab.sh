#!/bin/bash
# Instruction mix of the search kernel under the default library and under libtls_amd_<NAME>.so (one PMC pass each); via gpurun.
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
CFG=${2:-k2_90d}
OUT=$ROOT/gpurun_out/pmc_ab_$1
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for v in default $1; do
  if [ $v != default ]; then export TLS_AMD_DEBUG=1 TLS_AMD_LIB=$ROOT/tls_amd/libtls_amd_$v.so; fi
  timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS --output-format csv -d "$OUT/$v" -o k -- python $ROOT/tools/gpu_ab_time.py $CFG 1 > "$OUT/$v.log" 2>&1
  timeout 300 rocprofv3 --pmc SQ_INSTS_BRANCH SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_INSTS_VMEM --output-format csv -d "$OUT/${v}_b" -o k -- python $ROOT/tools/gpu_ab_time.py $CFG 1 > "$OUT/${v}_b.log" 2>&1
done
python - <<PY
import csv, collections, glob
for v in ("default", "$1"):
    for d in (v, v + "_b"):
        for f in glob.glob("$OUT/%s/*counter_collection.csv" % d):
            agg = collections.defaultdict(list)
            for r in csv.DictReader(open(f)):
                if "tls_search" in r["Kernel_Name"]:
                    agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
            for k, vals in sorted(agg.items()):
                print("%-10s %-28s %.5g  (%d launches)" % (v, k, sum(vals)/len(vals), len(vals)))
PY
