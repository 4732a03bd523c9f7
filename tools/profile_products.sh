#!/bin/bash
# profile_products.sh TAG: FETCH_SIZE / WRITE_SIZE passes over tools/pmc_products.py (own runs each), summary in
# gpurun_out/prof_products_TAG/traffic.json
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r}
OUT=$R/gpurun_out/prof_products_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch" -- python3 "$R/tools/pmc_products.py" > "$OUT/fetch.log" 2>&1 || { echo "fetch pass failed"; tail -5 "$OUT/fetch.log"; exit 1; }
timeout -s KILL 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/write" -- python3 "$R/tools/pmc_products.py" > "$OUT/write.log" 2>&1 || { echo "write pass failed"; tail -5 "$OUT/write.log"; exit 1; }
F=$(find "$OUT/fetch" -name "*counter_collection.csv" | head -1)
W=$(find "$OUT/write" -name "*counter_collection.csv" | head -1)
python3 "$R/tools/pmc_products_traffic.py" "$F" "$W" "$OUT/fetch.log" "$OUT/traffic.json" || exit 1
rm -rf "$OUT/fetch" "$OUT/write"
