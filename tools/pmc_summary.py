#!/usr/bin/env python3
"""Per-kernel means of the counters collected by tools/pmc_passes.sh:
    python tools/pmc_summary.py gpurun_out/pmc_<tag> [kernel-substring ...]"""
import collections
import csv
import glob
import sys

root = sys.argv[1]
wanted = sys.argv[2:] or ["csr_spmv", "dot_kernel"]
for path in sorted(glob.glob(f"{root}/*/*/*_counter_collection.csv")):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(path)):
        agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    print("==", path.split("/")[-3])
    for kernel, counters in agg.items():
        if not any(w in kernel for w in wanted):
            continue
        print("  ", kernel[:90])
        for name, vals in sorted(counters.items()):
            print(f"      {name:42s} {sum(vals) / len(vals):.6g}  (n={len(vals)})")
