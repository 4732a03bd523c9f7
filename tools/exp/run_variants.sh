# run_variants.sh NAME...: for each tools/exp/liblsk_NAME.so, the sliced-pattern GPU tests, then (only
# if they pass) the product's time on the headline matrix.
for v in "$@"; do
  export LSK_LIBRARY=$PWD/tools/exp/liblsk_$v.so
  if ! timeout -k 10 200 python -u -m pytest tests/test_gpu_spmv.py tests/test_gpu_fuzz.py -x -q --timeout 120 --timeout-method thread -m gpu -k sliced > gpurun_out/variant_$v.log 2>&1; then
    echo "$v: sliced tests FAILED (gpurun_out/variant_$v.log); not timed"; exit 1
  fi
  for fuse in 1 0; do
  LSK_FUSE=$fuse LSK_TUNE_REDRAWS=0 timeout -k 10 200 python tools/sliced_rate.py 512 4 f64 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', 'fuse $fuse', 'spmv', d['sliced_ms'], 'cg step', d['cg_ms_per_step'], 'placement', d['placement'], 'diff vs c32', d['max_rel_diff_vs_c32'])" || exit 1
  done
done
