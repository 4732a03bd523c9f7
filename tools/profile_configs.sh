#!/bin/bash
# profile_configs.sh TAG [config ...]: rocprofv3 --kernel-trace --stats of tools/config_rates.py, one run per
# configuration (cg2d, bicgstab, gmres by default), results under gpurun_out/configs_TAG/:
#   <config>_kernel_stats.csv   per-kernel durations of that run (set-up trials, plain and instrumented steps)
#   <config>.json               the line config_rates.py printed inside that run
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r}
shift || true
CONFIGS=${*:-cg2d bicgstab gmres}
OUT=$R/gpurun_out/configs_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for C in $CONFIGS; do
    echo "profiling $C"
    timeout -s KILL 500 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_$C" -- python3 "$R/tools/config_rates.py" "$C" > "$OUT/$C.json" 2> "$OUT/$C.err" || { echo "$C failed"; tail -5 "$OUT/$C.err"; exit 1; }
    find "$OUT/stats_$C" -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} "$OUT/${C}_kernel_stats.csv"
    rm -rf "$OUT/stats_$C"
done
