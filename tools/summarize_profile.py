"""Turn gpurun_out/prof_<tag>/ (tools/profile_gpu.sh) into the committed summaries under
profiles/: kernel stats, per-launch PMC means for the search kernel, and
profiles/hbm_traffic.json (read by bench.py for roofline.traffic).

HBM bytes follow /opt/skills/guides/MI355X_MICROARCH.md "HBM": FETCH_SIZE and WRITE_SIZE are
collected in separate --pmc passes, are in KiB, and on gfx950 FETCH_SIZE counts half of the
bytes of wide coalesced reads, so the read side is doubled (upper bound for our mix of narrow
and wide reads); WRITE_SIZE is used as reported (uncalibrated per the guide)."""
import collections
import csv
import json
import os
import shutil
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", "prof_" + tag)
dst = os.path.join(root, "profiles")
os.makedirs(dst, exist_ok=True)

shutil.copy(os.path.join(src, "trace", "k2_kernel_stats.csv"),
            os.path.join(dst, "%s_k2_90d_kernel_stats.csv" % tag))
bench = json.load(open(os.path.join(src, "bench.json")))
json.dump(bench, open(os.path.join(dst, "%s_bench_k2_90d.json" % tag), "w"), indent=1)

means = {}
for name in ("pmc_fetch", "pmc_write", "pmc_sq"):
    path = os.path.join(src, name, "k2_counter_collection.csv")
    if not os.path.exists(path):
        continue
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if "tls_search_kernel" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        means[k] = {"launches": len(v), "mean": sum(v) / len(v), "min": min(v), "max": max(v)}

with open(os.path.join(dst, "%s_k2_90d_pmc_summary.csv" % tag), "w") as fh:
    fh.write("counter,launches,mean_per_launch,min,max\n")
    for k in sorted(means):
        m = means[k]
        fh.write("%s,%d,%.6g,%.6g,%.6g\n" % (k, m["launches"], m["mean"], m["min"], m["max"]))

if "FETCH_SIZE" in means and "WRITE_SIZE" in means:
    fetch_kib, write_kib = means["FETCH_SIZE"]["mean"], means["WRITE_SIZE"]["mean"]
    rec = {"config": "k2_90d", "n_periods": 9679, "tag": tag,
           "fetch_size_kib_raw": fetch_kib, "write_size_kib_raw": write_kib,
           "bytes_per_launch": (2.0 * fetch_kib + write_kib) * 1024.0,
           "bytes_per_launch_uncorrected": (fetch_kib + write_kib) * 1024.0,
           "note": "read side doubled per MI355X_MICROARCH.md (gfx950 FETCH_SIZE counts 64 B per "
                   "128 B request); separate --pmc passes of `bench.py --steps 20`"}
    json.dump(rec, open(os.path.join(dst, "hbm_traffic.json"), "w"), indent=1)
    print(rec)
print(open(os.path.join(dst, "%s_k2_90d_kernel_stats.csv" % tag)).read())
