cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for s in 1 0; do
  TLS_SCREEN32=$s rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $R/gpurun_out/freq_$s -o f -- python $R/tools/gpu_ab_time.py k2_90d 2 > $R/gpurun_out/freq_$s.log 2>&1
  python - <<PY
import csv, glob
cc=[float(r["Counter_Value"]) for f in glob.glob("$R/gpurun_out/freq_$s/*counter_collection.csv") for r in csv.DictReader(open(f)) if "tls_search" in r["Kernel_Name"]]
dur=[(float(r["End_Timestamp"])-float(r["Start_Timestamp"])) for f in glob.glob("$R/gpurun_out/freq_$s/*kernel_trace.csv") for r in csv.DictReader(open(f)) if "tls_search" in r["Kernel_Name"]]
import statistics
print("screen=$s launches", len(cc), len(dur), "GUI_ACTIVE/8 median %.4g cycles" % (statistics.median(cc)/8), "duration median %.1f us" % (statistics.median(dur)/1e3), "=> %.3f GHz" % (statistics.median(cc)/8/statistics.median(dur)))
PY
done
