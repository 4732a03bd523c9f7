#!/usr/bin/env python3
"""Experiment: do the solver's vectors, all allocated on 2 MiB boundaries, get in each other's way in HBM?  CG at 512^3
with every large vector started LSK_VECTOR_SKEW bytes further into its 2 MiB than the one before (0: as it is).
(The skew itself was a three-line patch of HipBackend.empty -- a leading offset of (count * skew) mod 2 MiB into an
allocation that much larger -- which is not in the tree: it made no difference.)"""
import os, sys, time
import torch
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
from legionsolvers_amd import Context, CGSolver  # noqa: E402
from legionsolvers_amd import benchmark_stencil as bs  # noqa: E402


def timed(fn, reps=20):
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
    for _ in range(10):
        solver.step()
    rates = [round(20 / bs.timed_steps(ctx, solver, 0, 20), 2) for _ in range(3)]
    vec = lambda i: planner.get_vector(i, 0)._own  # noqa: E731
    x, p, q, r = vec(0), vec(CGSolver.P), vec(CGSolver.Q), vec(CGSolver.R)
    one = torch.ones(1, dtype=torch.float64, device="cuda") * 1e-9
    partials = torch.empty(be.sqnorm_slots(r), dtype=torch.float64, device="cuda")
    t_xpay = timed(lambda: be.axpy_xpay(x, [one], p, [one], r))
    t_sq = timed(lambda: be.axpy_sqnorm(r, [one], q, partials))
    out = torch.empty(1, dtype=torch.float64, device="cuda")
    t_dot = timed(lambda: be.dot(p, q, out))
    entry = planner.row_partitioned_matrices[0]
    pv = planner.get_vector(CGSolver.P, 0)
    t_mv = timed(lambda: matrix.matvec(planner.get_vector(CGSolver.Q, 0), pv._own, pv.own_lo, pv.own_hi, entry.kernel_partition, 0))
    print(f"skew {os.environ.get('LSK_VECTOR_SKEW', '0'):>7}  CG it/s {rates}  placement {dict(matrix.sliced_placement)}  "
          f"product {t_mv:.3f} ms  axpy_xpay {t_xpay:.4f}  axpy_sqnorm {t_sq:.4f}  dot {t_dot:.4f}  "
          f"addresses mod 2 MiB {[v.data_ptr() % (2 << 20) for v in (x, p, q, r)]}", flush=True)


main()
