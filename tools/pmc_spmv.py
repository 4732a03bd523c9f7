#!/usr/bin/env python3
"""Workload for the rocprofv3 --pmc passes: a few CSR SpMV launches at the headline size plus
two streams with known byte counts to calibrate the counters against (a dot product reads
2*rows*8 bytes and writes nothing; an axpy reads 2*rows*8 and writes rows*8).

    rocprofv3 --pmc FETCH_SIZE --output-format csv -d out -- python3 tools/pmc_spmv.py
    rocprofv3 --pmc WRITE_SIZE --output-format csv -d out -- python3 tools/pmc_spmv.py
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from legionsolvers_amd import Context, IndexPartition, IndexSpace, PartitionedVector, Scalar  # noqa: E402
from legionsolvers_amd import stencil_generator as sg  # noqa: E402

n1 = int(os.environ.get("LSK_N", "512"))
ctx = Context(torch.device("cuda", 0))
_, offs, cfs = sg.benchmark_stencil(4)
rows = n1 ** 3
ip = IndexPartition.equal(IndexSpace(0, rows - 1), 1)
A = sg.create_linearized_csr_stencil_matrix(ctx, (n1, n1, n1), offs, cfs, ip)
kp = A.create_kernel_partition_from_range_partition(ip)
A.compress(0, rows - 1, kp, ip)  # the planner's set-up: 32-bit structure (unless LSK_COMPRESS=0) ...
A.tune(0, rows - 1, kp, ip)      # ... and the tile placement
A.slice_structure(0, rows - 1, kp, ip)  # ... and the sliced-pattern copy (unless LSK_SLICED=0)
x = PartitionedVector(ctx, "x", ip)
y = PartitionedVector(ctx, "y", ip)
x.owned.copy_(torch.randn(rows, device="cuda", dtype=torch.float64))
y.zero_fill()
held_sliced, A.sliced_window = A.sliced_window, None
if A.col32 is not None:  # the form that reads the 64-bit columns and rects, for comparison
    held, A.col32_window = A.col32_window, None
    for _ in range(3):
        A.matvec(y, x.owned, 0, rows - 1, kp, 0)
    A.col32_window = held
for _ in range(3):           # the 32-bit structure
    A.matvec(y, x.owned, 0, rows - 1, kp, 0)
A.sliced_window = held_sliced
if A.sliced:
    for _ in range(3):       # the sliced-pattern copy
        A.matvec(y, x.owned, 0, rows - 1, kp, 0)
a = Scalar(ctx, 0.5, torch.float64)
for _ in range(3):
    x.dot(y)
    y.axpy(a, x)
ctx.synchronize()
print(f"tile placement {A.placement} sliced {A.sliced_placement}")
if A.sliced:
    s = A.sliced[0]
    print(f"sliced_algorithmic_bytes={s.algorithmic_bytes(rows, 8)}")
print(f"rows={rows} nnz={A.nnz} spmv_algorithmic_bytes={A.nnz * 16 + rows * 32} "
      f"spmv_c32_algorithmic_bytes={A.nnz * 12 + rows * 20} "
      f"dot_bytes={rows * 16} axpy_read_bytes={rows * 16} axpy_write_bytes={rows * 8}")
