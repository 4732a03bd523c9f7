#!/usr/bin/env python3
"""Experiment: in whatever layout this process got, the 27-point 512^3 fp64 product for two sliced copies (the planner's
own, and one made behind a block of a third of the device) into output vectors in front of, between and behind them.
Run several times in one gpurun call: later processes on a box get layouts in which no place for the vectors alone
helps; does a second place for the COPY?"""
import os, sys
import torch
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
os.environ["LSK_PLACE_VECTORS"] = "0"
from legionsolvers_amd import Context  # noqa: E402
from legionsolvers_amd import benchmark_stencil as bs  # noqa: E402


def timed(fn, reps=5):
    fn(); fn()
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(reps):
        fn()
    t1.record()
    t1.synchronize()
    return t0.elapsed_time(t1) / reps


def main():
    ctx = Context(torch.device("cuda", 0))
    matrix, planner, solver = bs.build_problem(ctx, 4, 512, 512, 512, 1, 1, torch.float64, num_systems=1)
    be = ctx.backend
    rows = matrix.rows
    own = matrix.sliced[0]
    place = matrix.sliced_placement.get(0, 0)
    x = planner.get_vector(2, 0)._data
    ys = {"workspace": planner.get_vector(3, 0)._data}
    third = torch.cuda.mem_get_info()[1] // 3
    block = torch.empty(third, dtype=torch.uint8, device="cuda")
    ys["behind block"] = torch.zeros(rows, dtype=torch.float64, device="cuda")
    second = be.csr_sliced_create(rows, matrix.nnz, matrix.rects, int(matrix.k_base), matrix.col, matrix.val, 0, 0, rows - 1)
    ys["behind second copy"] = torch.zeros(rows, dtype=torch.float64, device="cuda")
    for name, c in (("own copy", own), ("copy behind block", second)):
        row = {k: round(min(timed(lambda: be.csr_sliced_spmv(c, 0, rows, x, y, 0, p)) for p in {place, 2, 64}), 3) for k, y in ys.items()}
        print(f"{name}: {row}", flush=True)
    del block


main()
