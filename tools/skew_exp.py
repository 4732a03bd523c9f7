#!/usr/bin/env python3
"""Does the CSR SpMV time depend on the relative placement of col and val?  Both arrays are carved
out of ONE allocation, val at a varying offset after col; same kernel, same process."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from legionsolvers_amd import Context, CSRMatrix, IndexPartition, IndexSpace, PartitionedVector  # noqa: E402
from legionsolvers_amd import stencil_generator as sg  # noqa: E402

n1 = int(os.environ.get("LSK_N", "512"))
ctx = Context(torch.device("cuda", 0))
_, offs, cfs = sg.benchmark_stencil(4)
rows = n1 ** 3
space = IndexSpace(0, rows - 1)
ip = IndexPartition.equal(space, 1)
A0 = sg.create_linearized_csr_stencil_matrix(ctx, (n1, n1, n1), offs, cfs, ip)
nnz = A0.nnz
x = PartitionedVector(ctx, "x", ip)
y = PartitionedVector(ctx, "y", ip)
x.owned.copy_(torch.randn(rows, device="cuda", dtype=torch.float64))
slack = 64 << 20
pool = torch.empty(nnz * 16 + slack, dtype=torch.uint8, device="cuda")
col = pool[:nnz * 8].view(torch.int64)
col.copy_(A0.col)


def timed(fn, reps):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    b.synchronize()
    return a.elapsed_time(b) / reps


print(f"pool at {pool.data_ptr():#x}", flush=True)
for delta in [0, 256, 1024, 4096, 8192, 65536, 1 << 20, (1 << 20) + 4096, 3 << 20, 17 << 20, 0, 4096]:
    val = pool[nnz * 8 + delta:nnz * 8 + delta + nnz * 8].view(torch.float64)
    val.copy_(A0.val)
    A = CSRMatrix(ctx, A0.rects, 0, col, val, space, 0)
    kp = A.create_kernel_partition_from_range_partition(ip)

    def spmv():
        A.matvec(y, x.owned, 0, rows - 1, kp, 0)

    for _ in range(60):
        spmv()
    torch.cuda.synchronize()
    ts = [timed(spmv, 10) for _ in range(3)]
    print(f"delta {delta:>10d}  (col-val offset mod 1MiB = {((nnz * 8 + delta) % (1 << 20)):>8d})  ms " +
          " ".join(f"{t:.2f}" for t in ts), flush=True)
