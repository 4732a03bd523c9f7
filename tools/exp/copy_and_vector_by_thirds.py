#!/usr/bin/env python3
"""Experiment (first process on a fresh device, where memory is handed out from the bottom up): three sliced-pattern
copies of the 27-point 512^3 fp64 matrix, one in each 103 GB part of HBM (pads sized from the allocations so far), one
output vector in each part, x in the first: the product's time for every (copy, y) pair, plus a fourth copy made to
straddle the boundary between the first two parts."""
import os, sys
import torch
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
os.environ["LSK_PLACE_VECTORS"] = "0"
from legionsolvers_amd import Context, IndexPartition, IndexSpace  # noqa: E402
from legionsolvers_amd import stencil_generator as sg  # noqa: E402

GB = 10 ** 9


def timed(fn, reps=5):
    fn(); fn()
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(reps):
        fn()
    t1.record()
    t1.synchronize()
    return t0.elapsed_time(t1) / reps


def used():
    free, total = torch.cuda.mem_get_info()
    return (total - free) / GB


def main():
    straddle = len(sys.argv) > 1 and sys.argv[1] == "straddle"
    ctx = Context(torch.device("cuda", 0))
    be = ctx.backend
    n1 = 512
    rows = n1 ** 3
    ip = IndexPartition.equal(IndexSpace(0, rows - 1), 1)
    _, offs, cfs = sg.benchmark_stencil(4)
    A = sg.create_linearized_csr_stencil_matrix(ctx, (n1, n1, n1), offs, cfs, ip)
    third = torch.cuda.mem_get_info()[1] / 3 / GB
    g = torch.Generator(device="cuda").manual_seed(2)
    x = torch.randn(rows, dtype=torch.float64, device="cuda", generator=g)
    make = lambda: be.csr_sliced_create(rows, A.nnz, A.rects, int(A.k_base), A.col, A.val, 0, 0, rows - 1)  # noqa: E731
    ys, copies, where = [], [], []
    pads = []
    print(f"a third of the device: {third:.1f} GB; in use before the first copy {used():.1f} GB", flush=True)
    if straddle:   # one copy across the boundary between the first two parts, y on either side of it and in the third part
        ys.append(torch.zeros(rows, dtype=torch.float64, device="cuda")); where.append(f"y at {used():.0f} GB")
        pads.append(torch.empty(int((third - 15 - used()) * GB), dtype=torch.uint8, device="cuda"))
        at = used(); copies.append(make()); where.append(f"copy at {at:.0f}-{used():.0f} GB")
        ys.append(torch.zeros(rows, dtype=torch.float64, device="cuda")); where.append(f"y at {used():.0f} GB")
        pads.append(torch.empty(int((2 * third + 1 - used()) * GB), dtype=torch.uint8, device="cuda"))
        ys.append(torch.zeros(rows, dtype=torch.float64, device="cuda")); where.append(f"y at {used():.0f} GB")
    else:
        for part in range(3):
            if used() < part * third + 1:
                pads.append(torch.empty(int((part * third + 1 - used()) * GB), dtype=torch.uint8, device="cuda"))
            ys.append(torch.zeros(rows, dtype=torch.float64, device="cuda")); where.append(f"y{part} at {used():.0f} GB")
            at = used(); copies.append(make()); where.append(f"copy{part} at {at:.0f}-{used():.0f} GB")
    print("; ".join(where), flush=True)
    for rnd in range(2):
        for ci, c in enumerate(copies):
            row = []
            for y in ys:
                best = min(timed(lambda: be.csr_sliced_spmv(c, 0, rows, x, y, 0, place)) for place in (2, 64))
                row.append(round(best, 3))
            print(f"round {rnd} copy {ci}: product ms into each y (best of two workgroup placements): {row}", flush=True)


main()
