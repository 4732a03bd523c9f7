#!/usr/bin/env python3
"""Host cost of one CG step: the solver on a grid so small (32^3) that the GPU finishes a step's
kernels faster than the host issues them, so wall time per step = what the host spends enqueueing.
Prints the wall time per step and, with --profile, where it goes (cProfile, top functions)."""
import argparse
import cProfile
import os
import pstats
import sys
import time

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

from legionsolvers_amd import Context  # noqa: E402
from legionsolvers_amd import benchmark_stencil as bs  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=32)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--solver", type=int, default=1)
    ap.add_argument("--pieces", type=int, default=1)
    ap.add_argument("--profile", action="store_true")
    args = ap.parse_args()
    ctx = Context(torch.device("cuda", 0))
    _, planner, solver = bs.build_problem(ctx, 4, args.n, args.n, args.n, args.pieces, args.solver, torch.float64,
                                          num_systems=1)
    for _ in range(20):
        solver.step()
    ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        solver.step()
    t_issue = time.perf_counter() - t0
    ctx.synchronize()
    t_all = time.perf_counter() - t0
    print(f"{args.n}^3 solver {args.solver} pieces {args.pieces}: issue {t_issue / args.steps * 1e3:.4f} ms/step, "
          f"with the final sync {t_all / args.steps * 1e3:.4f} ms/step", flush=True)
    if args.profile:
        prof = cProfile.Profile()
        prof.enable()
        for _ in range(args.steps):
            solver.step()
        prof.disable()
        ctx.synchronize()
        pstats.Stats(prof).sort_stats("tottime").print_stats(22)


if __name__ == "__main__":
    main()
