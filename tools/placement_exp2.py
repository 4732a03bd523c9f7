#!/usr/bin/env python3
"""Which array's physical placement moves the CSR SpMV time?  Phase 1 re-allocates only x and y
around a fixed matrix; phase 2 re-allocates the matrix around fixed vectors."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from legionsolvers_amd import Context, IndexPartition, IndexSpace, PartitionedVector  # noqa: E402
from legionsolvers_amd import stencil_generator as sg  # noqa: E402

n1 = 512
ctx = Context(torch.device("cuda", 0))
_, offs, cfs = sg.benchmark_stencil(4)
rows = n1 ** 3
ip = IndexPartition.equal(IndexSpace(0, rows - 1), 1)


def timed(fn, reps):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    b.synchronize()
    return a.elapsed_time(b) / reps


def measure(A, kp, x, y):
    def spmv():
        A.matvec(y, x.owned, 0, rows - 1, kp, 0)
    for _ in range(60):
        spmv()
    torch.cuda.synchronize()
    return min(timed(spmv, 10) for _ in range(3))


A = sg.create_linearized_csr_stencil_matrix(ctx, (n1, n1, n1), offs, cfs, ip)
kp = A.create_kernel_partition_from_range_partition(ip)
hold = []
for trial in range(6):
    x = PartitionedVector(ctx, "x", ip)
    y = PartitionedVector(ctx, "y", ip)
    x.owned.copy_(torch.randn(rows, device="cuda", dtype=torch.float64))
    print(f"phase 1 (new x, y) trial {trial}: {measure(A, kp, x, y):.2f} ms  x={x.owned.data_ptr():#x}", flush=True)
    hold.append((x, y))  # keep them, so that the next pair lands elsewhere
x, y = hold[0]
del hold
torch.cuda.empty_cache()
for trial in range(5):
    del A, kp
    torch.cuda.empty_cache()
    pad = torch.empty((trial * 37 + 1) << 20, dtype=torch.uint8, device="cuda")
    A = sg.create_linearized_csr_stencil_matrix(ctx, (n1, n1, n1), offs, cfs, ip)
    kp = A.create_kernel_partition_from_range_partition(ip)
    print(f"phase 2 (new matrix) trial {trial}: {measure(A, kp, x, y):.2f} ms  col={A.col.data_ptr():#x}", flush=True)
    del pad
