#!/usr/bin/env python3
"""Experiment: does the time of one product into a vector behind 0, 1 or 2 blocks of a third of the device predict
the time of a whole CG-like step on three vectors there (product, two dots, three updates)?  Run with
LSK_PLACE_VECTORS=0 so the planner's own workspace stays in front; several runs in one call show what later
processes on a box see."""
import os, sys
import torch
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
os.environ["LSK_PLACE_VECTORS"] = "0"
from legionsolvers_amd import Context  # noqa: E402
from legionsolvers_amd import benchmark_stencil as bs  # noqa: E402


def timed(fn, reps):
    fn()
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(reps):
        fn()
    t1.record()
    t1.synchronize()
    return t0.elapsed_time(t1) / reps


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    ctx = Context(torch.device("cuda", 0))
    matrix, planner, solver = bs.build_problem(ctx, 4, n, n, n, 1, 1, torch.float64, num_systems=1)
    be = ctx.backend
    rows = matrix.rows
    sol = planner.sol_vectors[0]._data
    handle, place = matrix.sliced[0], matrix.sliced_placement.get(0, 0)
    own = bs.timed_steps(ctx, solver, 5, 40)
    print(f"solver own step (workspace in front): {own * 1e3 / 40:.3f} ms", flush=True)
    third = torch.cuda.mem_get_info()[1] // 3
    alpha = be.new_scalar(1e-3, torch.float64)
    out = torch.zeros(1, dtype=torch.float64, device="cuda")
    kept = []
    for depth in range(3):
        if depth:
            if torch.cuda.mem_get_info()[0] < third + (8 << 30):
                break
            kept.append(torch.empty(third, dtype=torch.uint8, device="cuda"))
        p, q, r = (torch.zeros(rows, dtype=torch.float64, device="cuda") for _ in range(3))
        kept += [p, q, r]

        def product():
            be.csr_sliced_spmv(handle, 0, rows, p, q, 0, place)

        def step():
            be.csr_sliced_spmv(handle, 0, rows, p, q, 0, place)
            be.dot(p, q, out)
            be.axpy(sol, [alpha], p)
            be.axpy(r, [alpha], q)
            be.dot(r, r, out)
            be.xpay(p, [alpha], r)

        def updates():
            be.dot(p, q, out)
            be.axpy(sol, [alpha], p)
            be.axpy(r, [alpha], q)
            be.dot(r, r, out)
            be.xpay(p, [alpha], r)
        print(f"depth {depth}: product {timed(product, 5):.3f} ms  updates {timed(updates, 5):.3f} ms  step {timed(step, 10):.3f} ms", flush=True)


main()
