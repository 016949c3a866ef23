#!/bin/bash
# Developer tool: issue mix of the search kernel with the fp32 screen on and off (rocprofv3 --pmc, own passes).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for s in 1 0; do
  for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_INSTS_VALU_FMA_F64 SQ_INSTS_BRANCH" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS"; do
    rm -rf $R/gpurun_out/scrpmc; TLS_SCREEN32=$s rocprofv3 --pmc $set --output-format csv -d $R/gpurun_out/scrpmc -o f -- python $R/tools/gpu_ab_time.py k2_90d 1 > /dev/null 2>&1
    python - <<PY
import csv, glob, collections
agg=collections.defaultdict(list)
for f in glob.glob("$R/gpurun_out/scrpmc/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "tls_search" in r["Kernel_Name"]: agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
print("screen=$s", " ".join("%s=%.4g" % (k, sum(v)/len(v)) for k,v in sorted(agg.items())))
PY
  done
done
