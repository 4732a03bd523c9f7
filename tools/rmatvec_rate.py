#!/usr/bin/env python3
"""Time of the transposed product y <- A^T x beside y <- A x on the headline matrix (27-point
stencil, fp64, one GPU); at that size, the adjoint identity <A u, v> = <u, A^T v> and, the matrix
being symmetric, A^T v against A v element by element.

    python tools/rmatvec_rate.py [n]      (grid n^3, default 512)

Prints one JSON line."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from legionsolvers_amd import Context  # noqa: E402
from legionsolvers_amd import benchmark_stencil as bs  # noqa: E402


def timed(ctx, fn, reps):
    fn()
    ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    ctx.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    ctx = Context(torch.device("cuda", 0))
    matrix, planner, _ = bs.build_problem(ctx, 4, n, n, n, 1, 1, torch.float64, num_systems=1)
    # CG's workspace: vectors 2, 3, 4
    planner.get_vector(2, 0).random_fill(1)
    planner.get_vector(3, 0).random_fill(2)
    t0 = time.perf_counter()
    planner.rmatvec(4, 3)                      # w = A^T v (builds the transpose)
    ctx.synchronize()
    setup_s = time.perf_counter() - t0
    lhs_r = planner.dot(2, 4).get_value()      # <u, A^T v>
    atv = planner.get_vector(4, 0).owned.clone()
    planner.matvec(4, 3)                       # w = A v: the same vector, summed in another order
    av = planner.get_vector(4, 0).owned
    sym = float((atv - av).abs().max() / av.abs().max())
    del atv
    planner.matvec(4, 2)                       # w = A u
    lhs = planner.dot(4, 3).get_value()        # <A u, v>
    t_r = timed(ctx, lambda: planner.rmatvec(4, 3), 20)
    t_a = timed(ctx, lambda: planner.matvec(4, 2), 20)
    tb = matrix._transposed
    print(json.dumps({"grid": n, "rows": matrix.rows, "nnz": matrix.nnz, "matvec_ms": round(t_a, 3),
                      "rmatvec_ms": round(t_r, 3), "transpose_setup_s": round(setup_s, 2),
                      "runs": len(tb.parts), "u_ATv": lhs_r, "Au_v": lhs,
                      "adjoint_rel_diff": abs(lhs - lhs_r) / abs(lhs), "ATv_vs_Av_max_rel_diff": sym}), flush=True)


if __name__ == "__main__":
    main()
