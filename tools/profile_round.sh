#!/bin/bash
# profile_round.sh TAG: the round's profiling runs on the GPU box, results under gpurun_out/prof_TAG/:
#   stats/      rocprofv3 --kernel-trace --stats of a short bench.py run (per-kernel durations)
#   fetch/ write/  rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over tools/pmc_spmv.py (own runs each)
#   traffic.json   HBM bytes per SpMV launch from those two passes (tools/pmc_traffic.py)
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
export LSK_BENCH_HOST_COST=0   # (its 32^3 launches would share kernel names with the timed ones)
timeout -s KILL 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 "$R/bench.py" --steps 20 --warmup 5 --no-cpu-baseline > "$OUT/bench_under_rocprof.json" 2> "$OUT/bench_under_rocprof.err" || { echo "stats pass failed"; tail -5 "$OUT/bench_under_rocprof.err"; exit 1; }
export LSK_TUNE_REDRAWS=0
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch" -- python3 "$R/tools/pmc_spmv.py" > "$OUT/fetch.log" 2>&1 || { echo "fetch pass failed"; tail -5 "$OUT/fetch.log"; exit 1; }
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/write" -- python3 "$R/tools/pmc_spmv.py" > "$OUT/write.log" 2>&1 || { echo "write pass failed"; tail -5 "$OUT/write.log"; exit 1; }
F=$(find "$OUT/fetch" -name "*counter_collection.csv" | head -1)
W=$(find "$OUT/write" -name "*counter_collection.csv" | head -1)
S=$(grep -o "sliced_algorithmic_bytes=[0-9]*" "$OUT/fetch.log" | cut -d= -f2)
P=$(grep "tile placement" "$OUT/fetch.log" | head -1)
python3 "$R/tools/pmc_traffic.py" "$F" "$W" "$OUT/traffic.json" "$P" "$S"
find "$OUT/stats" -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} "$OUT/kernel_stats.csv"
# the counter CSVs are large: keep only the per-kernel means
rm -rf "$OUT/fetch" "$OUT/write" "$OUT/stats"
