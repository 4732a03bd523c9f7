#!/usr/bin/env python3
"""Workload for a rocprofv3 --pmc FETCH_SIZE pass over the tile -> XCD placements of the CSR SpMV
(32-bit structure, headline matrix): three launches per placement, in the order printed, so the
per-dispatch counter rows of the product kernel can be matched to placements by position; with
LSK_PMC_SUMMARISE=<counter_collection.csv> it prints the mean per placement instead.

    rocprofv3 --pmc FETCH_SIZE --output-format csv -d out -- python3 tools/pmc_placements.py
    LSK_PMC_SUMMARISE=out/.../counter_collection.csv python3 tools/pmc_placements.py
"""
import csv
import json
import os
import sys

PLACEMENTS = [0, 16, 64, 256, 1024, 4096]
REPS = 3

if os.environ.get("LSK_PMC_SUMMARISE"):
    rows = [r for r in csv.DictReader(open(os.environ["LSK_PMC_SUMMARISE"]))
            if r["Counter_Name"] == "FETCH_SIZE" and "csr_spmv" in r["Kernel_Name"] and "unsigned int" in r["Kernel_Name"]]
    vals = [float(r["Counter_Value"]) for r in rows][-len(PLACEMENTS) * REPS:]
    out = {}
    for i, p in enumerate(PLACEMENTS):
        kb = sum(vals[i * REPS:(i + 1) * REPS]) / REPS
        out[str(p)] = {"FETCH_SIZE_KB_mean": kb, "fetch_bytes_x2": 2 * kb * 1024}
    print(json.dumps(out, indent=1))
    sys.exit(0)

import torch  # noqa: E402

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from legionsolvers_amd import Context, IndexPartition, IndexSpace, PartitionedVector  # noqa: E402
from legionsolvers_amd import stencil_generator as sg  # noqa: E402

n1 = int(os.environ.get("LSK_N", "512"))
ctx = Context(torch.device("cuda", 0))
_, offs, cfs = sg.benchmark_stencil(4)
rows = n1 ** 3
ip = IndexPartition.equal(IndexSpace(0, rows - 1), 1)
A = sg.create_linearized_csr_stencil_matrix(ctx, (n1, n1, n1), offs, cfs, ip)
kp = A.create_kernel_partition_from_range_partition(ip)
assert A.compress(0, rows - 1, kp, ip)
x = PartitionedVector(ctx, "x", ip)
y = PartitionedVector(ctx, "y", ip)
x.random_fill(1)
y.zero_fill()
A.placement = {0: 0}
A.matvec(y, x.owned, 0, rows - 1, kp, 0)
ctx.synchronize()
for p in PLACEMENTS:
    A.placement = {0: p}
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(REPS):
        A.matvec(y, x.owned, 0, rows - 1, kp, 0)
    t1.record()
    t1.synchronize()
    print(f"placement {p}: {t0.elapsed_time(t1) / REPS:.3f} ms per launch", flush=True)
