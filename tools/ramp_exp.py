#!/usr/bin/env python3
"""SpMV launch time as a function of how long the GPU has been busy (clock / fabric ramp)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from legionsolvers_amd import Context, IndexPartition, IndexSpace, PartitionedVector  # noqa: E402
from legionsolvers_amd import stencil_generator as sg  # noqa: E402

n1 = 512
ctx = Context(torch.device("cuda", 0))
_, offs, cfs = sg.benchmark_stencil(4)
rows = n1 ** 3
ip = IndexPartition.equal(IndexSpace(0, rows - 1), 1)
A = sg.create_linearized_csr_stencil_matrix(ctx, (n1, n1, n1), offs, cfs, ip)
kp = A.create_kernel_partition_from_range_partition(ip)
x = PartitionedVector(ctx, "x", ip)
y = PartitionedVector(ctx, "y", ip)
x.owned.copy_(torch.randn(rows, device="cuda", dtype=torch.float64))
torch.cuda.synchronize()
t_start = time.perf_counter()
secs = float(os.environ.get("RAMP_SECONDS", "40"))
while time.perf_counter() - t_start < secs:
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(50):
        A.matvec(y, x.owned, 0, rows - 1, kp, 0)
    b.record()
    b.synchronize()
    print(f"t={time.perf_counter() - t_start:6.2f}s  {a.elapsed_time(b) / 50:.3f} ms", flush=True)
