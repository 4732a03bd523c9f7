#!/bin/bash
# run_pair_variants.sh NAME...: tools/sliced_rate.py (the two-vector forms' launch times) on each variant
# library tools/exp/liblsk_NAME.so (build_variant_of.sh sliced_spmv NAME -D...; "main" = the package's).
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for n in "$@"; do
  if [ "$n" = main ]; then unset LSK_LIBRARY; else export LSK_LIBRARY=$R/tools/exp/liblsk_$n.so; fi
  timeout -k 10 200 python tools/sliced_rate.py 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('$n', {k: v for k, v in d.items() if k.startswith('pair') or k in ('sliced_ms', 'wave_dma_form_ms', 'wave_dma_form_dot_w_is_x_ms')})
" || exit 1
done
