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
        if "tls_search_kernel" in r["Kernel_Name"] or "tls_slim_kernel" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        means[k] = {"launches": len(v), "mean": sum(v) / len(v), "min": min(v), "max": max(v)}

with open(os.path.join(dst, "%s_k2_90d_pmc_summary.csv" % tag), "w") as fh:
    fh.write("counter,launches,mean_per_launch,min,max\n")
    for k in sorted(means):
        m = means[k]
        fh.write("%s,%d,%.6g,%.6g,%.6g\n" % (k, m["launches"], m["mean"], m["min"], m["max"]))

def update_traffic_record(rec):
    """profiles/hbm_traffic.json keeps one record per configuration, labelled with the commit it was measured at."""
    path = os.path.join(dst, "hbm_traffic.json")
    doc = json.load(open(path)) if os.path.exists(path) else {}
    recs = [r for r in doc.get("records", []) if r.get("config") != rec["config"]]
    recs.append(rec)
    doc["records"] = recs
    doc.setdefault("note", "read side doubled per MI355X_MICROARCH.md (gfx950 FETCH_SIZE counts 64 B per 128 B request); "
                           "separate --pmc passes; bench.py labels roofline.traffic with the commit of the record")
    json.dump(doc, open(path, "w"), indent=1)


commit = os.environ.get("TLS_COMMIT", "unknown")
sys.path.insert(0, root)
import bench as _bench   # kernel_sources_digest: which sources the counters were taken on (bench.py compares)
sources = _bench.kernel_sources_digest()
if "FETCH_SIZE" in means and "WRITE_SIZE" in means:
    fetch_kib, write_kib = means["FETCH_SIZE"]["mean"], means["WRITE_SIZE"]["mean"]
    rec = {"config": "k2_90d", "n_periods": 9679, "commit": commit, "kernel_sources": sources,
           "source": "FETCH_SIZE x2 + WRITE_SIZE, profiles/%s_k2_90d_pmc_summary.csv" % tag,
           "fetch_size_kib_raw": fetch_kib, "write_size_kib_raw": write_kib,
           "bytes_per_launch": (2.0 * fetch_kib + write_kib) * 1024.0}
    update_traffic_record(rec)
    print(rec)
kep = os.path.join(root, "gpurun_out", "prof_kepler_" + tag + ".json")
if os.path.exists(kep):
    k = json.load(open(kep))
    for variant in k:
        rec = {"config": "kepler_4yr" if variant["name"] == "default" else "kepler_4yr/" + variant["name"], "n_periods": variant["n_periods"],
               "commit": commit, "kernel_sources": sources, "source": "FETCH_SIZE x2 + WRITE_SIZE, profiles/%s_kepler_hbm_traffic.json (every 64th period)" % tag,
               "fetch_size_kib_raw": variant["fetch_kib"], "write_size_kib_raw": variant["write_kib"],
               "bytes_per_launch": (2.0 * variant["fetch_kib"] + variant["write_kib"]) * 1024.0, "kernel_avg_ms": variant["kernel_ms"]}
        update_traffic_record(rec)
    json.dump(k, open(os.path.join(dst, "%s_kepler_hbm_traffic.json" % tag), "w"), indent=1)
    print(k)
print(open(os.path.join(dst, "%s_k2_90d_kernel_stats.csv" % tag)).read())
