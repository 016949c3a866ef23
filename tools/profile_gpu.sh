#!/bin/bash
# Run on the GPU box (via gpurun): rocprofv3 kernel trace + HBM counter passes of bench.py.
# Outputs land in gpurun_out/prof_$TAG/; tools/summarize_profile.py turns them into profiles/.
set -u
TAG=${1:-r02}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
# the kernel trace runs the SAME command the bench line below comes from (default steps and warm-up); the counter passes a shorter one
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o k2 -- python $ROOT/bench.py --no-cpu-baseline --no-extras > "$OUT/trace.log" 2>&1
BENCH="python $ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras"
# counters in their own passes (TCC slots: FETCH_SIZE 3, WRITE_SIZE 2), no trace domains besides kernels
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -o k2 -- $BENCH > "$OUT/pmc_fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -o k2 -- $BENCH > "$OUT/pmc_write.log" 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY --output-format csv -d "$OUT/pmc_sq" -o k2 -- $BENCH > "$OUT/pmc_sq.log" 2>&1
python $ROOT/bench.py > "$OUT/bench.json" 2> "$OUT/bench.err"
ls -R "$OUT" | head -40
