#!/bin/bash
# After tools/gpu_round.sh <tag>, tools/profile_pmc2.sh <tag>b and tools/profile_config.sh tess_27d 1 <tag> have run on the
# GPU box (gpurun merges gpurun_out/ back): copy the summaries the judge reads into profiles/<tag>_*.
set -eu
TAG=${1:-r04}
cd "$(dirname "$0")/.."
COMMIT=$(git rev-parse --short=8 HEAD)
TLS_COMMIT=$COMMIT python tools/summarize_profile.py $TAG > /dev/null
cp gpurun_out/round_$TAG/phases.txt profiles/${TAG}_phase_cycles.txt
if [ -f gpurun_out/round_$TAG/pytest_debug.txt ]; then cp gpurun_out/round_$TAG/pytest_debug.txt profiles/${TAG}_debug_checked_run.txt; fi
if [ -f gpurun_out/profile_post_$TAG.log ]; then grep -v "rocprofv3\]\|output_stream\|simple_timer" gpurun_out/profile_post_$TAG.log > profiles/${TAG}_post_search_kernels.txt; fi
cp gpurun_out/prof_tess_27d_$TAG/trace/k_kernel_stats.csv profiles/${TAG}_tess_kernel_stats.csv
cp gpurun_out/prof_kepler_${TAG}_default/trace/k_kernel_stats.csv profiles/${TAG}_kepler_sample_kernel_stats.csv
if [ -f gpurun_out/prof_kepler_${TAG}_full/trace/k_kernel_stats.csv ]; then cp gpurun_out/prof_kepler_${TAG}_full/trace/k_kernel_stats.csv profiles/${TAG}_kepler_full_grid_kernel_stats.csv; fi
python - <<PY
import csv, glob, collections, json
commit, tag = "$COMMIT", "$TAG"
rows = []
for d in ("p1", "p2", "p3", "p4"):
    for f in glob.glob("gpurun_out/prof_%sb/%s/*counter_collection.csv" % (tag, d)):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "tls_search_kernel<true, true, unsigned short, false, false, false>" in r["Kernel_Name"] or "tls_slim_kernel<false, 256>" in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in sorted(agg.items()):
            rows.append((k, sum(v) / len(v), len(v)))
with open("profiles/%s_k2_90d_pmc_issue_mix.csv" % tag, "w") as fh:
    fh.write("# rocprofv3 --pmc passes (four separate runs, kernel-trace only) of bench.py --steps 10 --warmup 2 --no-cpu-baseline "
             "--no-extras (tools/profile_pmc2.sh): mean per launch of the plain search kernel "
             "tls_slim_kernel<false, 256> (the four-slot kernel config 2 takes), commit %s\n" % commit)
    fh.write("counter,mean_per_launch,launches\n")
    for k, m, n in rows:
        fh.write("%s,%.6g,%d\n" % (k, m, n))
t = json.load(open("gpurun_out/prof_tess_27d_%s.json" % tag)); t["commit"] = commit
json.dump(t, open("profiles/%s_tess_hbm_traffic.json" % tag, "w"))
doc = json.load(open("profiles/hbm_traffic.json"))
recs = [r for r in doc["records"] if r.get("config") != "tess_27d"]
import sys; sys.path.insert(0, ".")
import bench as _bench
recs.append({"config": "tess_27d", "n_periods": 2459, "commit": commit, "kernel_sources": _bench.kernel_sources_digest(),
             "source": "FETCH_SIZE x2 + WRITE_SIZE, profiles/%s_tess_hbm_traffic.json" % tag,
             "fetch_size_kib_raw": t["fetch_kib"], "write_size_kib_raw": t["write_kib"],
             "bytes_per_launch": t["bytes_per_launch"], "kernel_avg_ms": t["kernel_ms"]})
doc["records"] = recs
json.dump(doc, open("profiles/hbm_traffic.json", "w"), indent=1)
b = json.load(open("profiles/%s_bench_k2_90d.json" % tag))
print("cfg2 ms/step %.4f value %.4g frac %.4f kernel %.4f" % (b["ms_per_step"], b["value"], b["roofline"]["frac"], b["roofline"]["kernel_ms"]))
print("one_shot %.3f cold %.3f power %.2f (tess %s, kepler %s)" % (b["config"]["one_shot"]["ms"], b["config"]["one_shot"]["cold_ms"], b["config"]["power_call_wall_ms_per_light_curve"], b["tess_27d"].get("power_call_wall_ms"), b["kepler_4yr"].get("power_call_wall_ms")))
print("noisy", json.dumps(b["config"]["noisy_variant"])[:420])
for k in ("tess_27d", "kepler_4yr"):
    o = b[k]; print(k, "%.3f ms frac %.4f traffic x%.2f" % (o["kernel_ms"], o["roofline"]["frac"], o["roofline"]["traffic"] / o["roofline"]["algorithmic_bytes_per_launch"]))
print("survey", b["survey_1024"]["curves_per_s"], b["survey_1024"]["curves_per_s_power"])
print([(x["config"], round(x["speedup_if_ranks_ran_the_blocks"], 2)) for x in b["shard_balance"]])
print("cpu", b["cpu_baseline"]["value"], b["cpu_baseline"]["fastmath"]["value"])
PY
