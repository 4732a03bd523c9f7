#!/usr/bin/env python3
"""Does the CSR SpMV time depend on where the arrays land?  Rebuilds the headline matrix several
times in one process behind dummy allocations of different sizes and times the same kernel."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from legionsolvers_amd import Context, IndexPartition, IndexSpace, PartitionedVector  # noqa: E402
from legionsolvers_amd import stencil_generator as sg  # noqa: E402

n1 = int(os.environ.get("LSK_N", "512"))
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


MiB = 1 << 20
for trial, pad in enumerate([0, 0, 1 * MiB, 3 * MiB, 7 * MiB + 4096, 64 * MiB, 0, 333 * MiB, 2 * MiB, 0]):
    torch.cuda.empty_cache()
    dummy = torch.empty(max(pad, 1), dtype=torch.uint8, device="cuda")
    A = sg.create_linearized_csr_stencil_matrix(ctx, (n1, n1, n1), offs, cfs, ip)
    kp = A.create_kernel_partition_from_range_partition(ip)
    x = PartitionedVector(ctx, "x", ip)
    y = PartitionedVector(ctx, "y", ip)
    x.owned.copy_(torch.randn(rows, device="cuda", dtype=torch.float64))

    def spmv():
        A.matvec(y, x.owned, 0, rows - 1, kp, 0)

    for _ in range(100):
        spmv()
    torch.cuda.synchronize()
    ts = [timed(spmv, 10) for _ in range(3)]
    ptrs = {k: t.data_ptr() for k, t in (("rects", A.rects), ("col", A.col), ("val", A.val), ("x", x.owned), ("y", y.owned))}
    print(f"trial {trial} pad {pad:>10d}  ms " + " ".join(f"{t:.2f}" for t in ts) + "  " +
          " ".join(f"{k}={v:#x}" for k, v in ptrs.items()), flush=True)
    del A, kp, x, y, dummy
