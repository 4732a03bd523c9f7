#!/bin/bash
# run_c32_variants.sh NAME...: bench.py with LSK_SLICED=0 on each variant library tools/exp/liblsk_NAME.so
# ("main" = the package's); prints the 32-bit-structure product's launch time and the CG rate.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
export LSK_SLICED=0 LSK_BENCH_HOST_COST=0
for n in "$@"; do
  if [ "$n" = main ]; then unset LSK_LIBRARY; else export LSK_LIBRARY=$R/tools/exp/liblsk_$n.so; fi
  timeout -k 10 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('$n', 'c32 ms', round(d['roofline']['ms_per_launch'], 3), d['roofline']['kernel'], 'CG it/s', round(d['value'], 1), flush=True)
" || exit 1
done
