#!/usr/bin/env python3
"""CG on the COO form of the headline matrix (27-point stencil, fp64, one GPU): the row-sorted COO
matrix gets the same sliced-pattern copy as the CSR form; LSK_SLICED=0 shows the scan kernel on the
32-bit index fields.

    python tools/coo_stencil_rate.py [n]
"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from legionsolvers_amd import (CGSolver, Context, IndexPartition, IndexSpace, PartitionedVector,  # noqa: E402
                               SquarePlanner)
from legionsolvers_amd import stencil_generator as sg  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
ctx = Context(torch.device("cuda", 0))
rows = n ** 3
ip = IndexPartition.equal(IndexSpace(0, rows - 1), 1)
_, offs, cfs = sg.benchmark_stencil(4)
t0 = time.perf_counter()
A = sg.create_linearized_coo_stencil_matrix(ctx, (n, n, n), offs, cfs, ip)
rhs = PartitionedVector(ctx, "rhs", ip)
rhs.constant_fill(1.0)
sol = PartitionedVector(ctx, "sol", ip)
sol.zero_fill()
pl = SquarePlanner(ctx)
pl.add_sol_vector(sol)
pl.add_rhs_vector(rhs)
pl.add_row_partitioned_matrix(A, 0, 0)
solver = CGSolver(pl)
ctx.synchronize()
setup = time.perf_counter() - t0
for _ in range(5):
    solver.step()
ctx.synchronize()
t0 = time.perf_counter()
for _ in range(30):
    solver.step()
ctx.synchronize()
ms = (time.perf_counter() - t0) / 30 * 1e3
print(json.dumps({"grid": n, "form": "coo", "sliced": bool(A.sliced), "setup_s": round(setup, 2),
                  "cg_ms_per_step": round(ms, 3), "cg_it_per_s": round(1e3 / ms, 1),
                  "residual": solver.residual_norm_squared[-1].get_value()}), flush=True)
