#!/usr/bin/env python3
"""CG steps per second at 512^3 with the sliced product's workgroup placement forced to each candidate in turn
(the solver's own vectors, the loop as bench.py times it): does the placement the tuner picks on temporary
vectors carry over to the loop?"""
import json, os, sys, time
import torch
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
from legionsolvers_amd import Context  # noqa: E402
from legionsolvers_amd import benchmark_stencil as bs  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    ctx = Context(torch.device("cuda", 0))
    matrix, planner, solver = bs.build_problem(ctx, 4, n, n, n, 1, 1, torch.float64, num_systems=1)
    out = {"grid": n, "set_up_tuner_chose": dict(matrix.sliced_placement)}
    for _ in range(10):
        solver.step()
    out["first_product_tuner_chose"] = dict(matrix.sliced_placement)
    for rnd in range(3):
        for place in (0, 2, 4, 16, 64):
            matrix.sliced_placement[0] = place
            out.setdefault(f"place{place}", []).append(round(bs.timed_steps(ctx, solver, 3, 20) and 20 / bs.timed_steps(ctx, solver, 0, 20), 2))
    for k, v in out.items():
        print(k, v, flush=True)


main()
