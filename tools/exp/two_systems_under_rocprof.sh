set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
export PYTHONPATH=$R
for place in 1 0; do
  export LSK_PLACE_VECTORS=$place
  OUT=$R/gpurun_out/two_place$place
  mkdir -p $OUT
  timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 -m legionsolvers_amd.benchmark_stencil -dim 4 -nx 512 -ny 512 -nz 512 -vp 1 -solver 1 -it 65 -pt 1 -systems 2 > $OUT/out.log 2>&1 || { echo failed $place; tail -5 $OUT/out.log; exit 1; }
  find $OUT/stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats.csv
  rm -rf $OUT/stats
done
