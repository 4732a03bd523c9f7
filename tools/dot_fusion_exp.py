#!/usr/bin/env python3
"""What the fused product + dot costs next to the plain product and the dot it replaces, on the
planner's default structure (32-bit), one GPU.  python tools/dot_fusion_exp.py [--n 512] [--dim 4]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from legionsolvers_amd import Context, IndexPartition, IndexSpace, PartitionedVector  # noqa: E402
from legionsolvers_amd import benchmark_stencil as bs  # noqa: E402
from legionsolvers_amd import stencil_generator as sg  # noqa: E402


def timed(fn, reps):
    for _ in range(3):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    b.synchronize()
    return a.elapsed_time(b) / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=512)
    ap.add_argument("--dim", type=int, default=4)
    ap.add_argument("--reps", type=int, default=20)
    args = ap.parse_args()
    ctx = Context(torch.device("cuda", 0))
    _, offs, cfs = sg.benchmark_stencil(args.dim)
    extent = bs.grid_extent(args.dim, args.n, args.n, args.n)
    rows = 1
    for e in extent:
        rows *= e
    ip = IndexPartition.equal(IndexSpace(0, rows - 1), 1)
    A = sg.create_linearized_csr_stencil_matrix(ctx, extent, offs, cfs, ip)
    kp = A.create_kernel_partition_from_range_partition(ip)
    A.compress(0, rows - 1, kp, ip)
    x = PartitionedVector(ctx, "x", ip)
    y = PartitionedVector(ctx, "y", ip)
    x.owned.copy_(torch.randn(rows, device="cuda", dtype=torch.float64))
    be = ctx.backend
    out = be.empty(1, torch.float64)
    for r in range(3):
        plain = timed(lambda: A.matvec(y, x.owned, 0, rows - 1, kp, 0), args.reps)
        fused = timed(lambda: A.matvec(y, x.owned, 0, rows - 1, kp, 0, dot_with=x), args.reps)
        dot = timed(lambda: be.dot(x.owned, y.owned, out), args.reps)
        print(f"round {r}: product {plain:.3f} ms, product + dot fused {fused:.3f} ms (+{fused - plain:.3f}), "
              f"separate dot {dot:.3f} ms", flush=True)


if __name__ == "__main__":
    main()
