#!/bin/bash
# profile_bicgstab.sh: rocprofv3 kernel statistics of tools/config_rates.py bicgstab (the random COO matrix of the
# fourth benchmark configuration) by default and with LSK_FUSE=0 (every leaf launched as asked for): which kernels
# a BiCGStab step is made of, and what the planner's fusions are worth to it.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for mode in default nofuse; do
  if [ $mode = nofuse ]; then export LSK_FUSE=0; else unset LSK_FUSE; fi
  rm -rf /tmp/pb_$mode
  timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pb_$mode -- python3 $R/tools/config_rates.py bicgstab > /tmp/pb_$mode.log 2>&1 || { echo failed $mode; tail -3 /tmp/pb_$mode.log; exit 1; }
  grep steps_per_s /tmp/pb_$mode.log | cut -c1-140
  python3 - $mode <<'PY'
import csv, glob, sys
f = glob.glob(f"/tmp/pb_{sys.argv[1]}/*/*kernel_stats.csv")[0]
rows = list(csv.DictReader(open(f)))
for r in rows[:14]:
    print("   ", r["Name"].split("::")[-1][:60], r["Calls"], f'{float(r["AverageNs"]) / 1e3:.1f} us', r["Percentage"])
PY
done
