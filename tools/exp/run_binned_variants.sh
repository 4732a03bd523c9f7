#!/bin/bash
# run_binned_variants.sh [-s SHAPE] NAME...: the binned COO tests, then the full-size timing under
# rocprofv3 (for the time of each pass), on each variant library tools/exp/liblsk_NAME.so (built with
# build_variant_of.sh coo_binned NAME ...; "main" = the library in the package).  Stops at the first failure.
R=${GRAFT_REPO_ROOT:-/root/repo}
SHAPE=0:0
TESTS=1
if [ "$1" = "-n" ]; then TESTS=0; shift; fi   # timing probes that compute something else: no tests
if [ "$1" = "-s" ]; then SHAPE=$2; shift 2; fi
export TMPDIR=/tmp LSK_EXP_SCAN=0
for n in "$@"; do
  if [ "$n" = main ]; then unset LSK_LIBRARY; else export LSK_LIBRARY=$R/tools/exp/liblsk_$n.so; fi
  echo "== $n $SHAPE"
  cd $R
  if [ $TESTS = 1 ]; then
    timeout -k 10 200 python -u -m pytest tests/test_gpu_coo_binned.py -x -q --timeout 120 --timeout-method thread > gpurun_out/binned_variant_$n.log 2>&1 || { echo "tests failed on $n"; tail -20 gpurun_out/binned_variant_$n.log; exit 1; }
  fi
  rm -rf /tmp/bprof_$n; cd /tmp
  timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/bprof_$n -- python3 $R/tools/coo_binned_exp.py ${DT:-f64} $SHAPE > /tmp/bprof_$n.log 2>&1 || { echo "timing failed on $n"; tail -5 /tmp/bprof_$n.log; exit 1; }
  grep binned /tmp/bprof_$n.log | sed 's/(.*padded [0-9.%]*)//; s/of the 32-bit.*//'
  python3 - "$n" <<'PY'
import csv, glob, sys
f = glob.glob(f"/tmp/bprof_{sys.argv[1]}/*/*kernel_stats.csv")[0]
for r in csv.DictReader(open(f)):
    if "binned_" in r["Name"]:
        print("   ", r["Name"].split("::")[-1][:30], f'{float(r["AverageNs"]) / 1e6:.3f} ms')
PY
done
