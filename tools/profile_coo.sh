#!/bin/bash
# profile_coo.sh TAG: the COO product's profiling runs on the GPU box, results under gpurun_out/coo_TAG/:
#   kernel_stats.csv   rocprofv3 --kernel-trace --stats of tools/coo_binned_exp.py (per-kernel durations)
#   exp.log            what that run printed (HIP-event times of the same launches)
#   traffic.json       HBM bytes per launch of each kernel from FETCH_SIZE / WRITE_SIZE passes (own runs each)
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r}
OUT=$R/gpurun_out/coo_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 "$R/tools/coo_binned_exp.py" f64 0:0 > "$OUT/exp.log" 2>&1 || { echo "stats pass failed"; tail -5 "$OUT/exp.log"; exit 1; }
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch" -- python3 "$R/tools/coo_binned_exp.py" f64 0:0 > "$OUT/fetch.log" 2>&1 || { echo "fetch pass failed"; tail -5 "$OUT/fetch.log"; exit 1; }
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/write" -- python3 "$R/tools/coo_binned_exp.py" f64 0:0 > "$OUT/write.log" 2>&1 || { echo "write pass failed"; tail -5 "$OUT/write.log"; exit 1; }
F=$(find "$OUT/fetch" -name "*counter_collection.csv" | head -1)
W=$(find "$OUT/write" -name "*counter_collection.csv" | head -1)
python3 "$R/tools/pmc_coo_traffic.py" "$F" "$W" "$OUT/fetch.log" "$OUT/traffic.json"
find "$OUT/stats" -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} "$OUT/kernel_stats.csv"
rm -rf "$OUT/fetch" "$OUT/write" "$OUT/stats"
grep -v "amdgpu.ids\|simple_timer" "$OUT/exp.log"
