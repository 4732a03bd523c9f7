#!/usr/bin/env python3
"""Solver step rates of the reference's other benchmark configurations on ONE GPU (the headline
configuration is bench.py's):

  cg2d      Test06CSRSolveCG: CG, 2-D 5-point Laplacian 4096^2, fp64 CSR, one shard
  bicgstab  BiCGStab on the random nonsymmetric row-sorted COO matrix (15 625 000 rows x 32), fp64
  gmres     GMRES(30) on the 3-D 7-point stencil 768^3, fp32 CSR (a step = one restart cycle:
            31 SpMV, 496 dot, 465 axpy, 30 scal, 1 xpay, 30 axpy into the solution)

    python tools/config_rates.py [cg2d] [bicgstab] [gmres] [gmres_classical]

(gmres_classical: the same cycle with GMRESSolver's classical Gram-Schmidt option, every inner product
of a sweep in one pass over w: different arithmetic, the reference's sequence is `gmres`.)

Prints one JSON line per configuration: steps/s, ms per step, algorithmic GB moved per step and
the fraction of the 8 TB/s HBM peak that is."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from legionsolvers_amd import (BiCGStabSolver, CGSolver, COOMatrix, Context, GMRESSolver,  # noqa: E402
                               IndexPartition, IndexSpace, PartitionedVector, SquarePlanner)
from legionsolvers_amd import stencil_generator as sg  # noqa: E402


def planner_for(ctx, matrix, ip, dtype):
    rhs = PartitionedVector(ctx, "rhs", ip, dtype)
    rhs.constant_fill(1.0)
    sol = PartitionedVector(ctx, "sol", ip, dtype)
    sol.zero_fill()
    pl = SquarePlanner(ctx, dtype)
    pl.add_sol_vector(sol)
    pl.add_rhs_vector(rhs)
    pl.add_row_partitioned_matrix(matrix, 0, 0)
    return pl


def rate(ctx, solver, warm, steps):
    for _ in range(warm):
        solver.step()
    ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        solver.step()
    ctx.synchronize()
    return (time.perf_counter() - t0) / steps


def main():
    which = sys.argv[1:] or ["cg2d", "bicgstab", "gmres"]
    ctx = Context(torch.device("cuda", 0))
    for name in which:
        if name == "cg2d":
            nx, s = 4096, 8
            rows = nx * nx
            ip = IndexPartition.equal(IndexSpace(0, rows - 1), 1)
            _, offs, cfs = sg.benchmark_stencil(2)
            A = sg.create_linearized_csr_stencil_matrix(ctx, (nx, nx), offs, cfs, ip)
            solver = CGSolver(planner_for(ctx, A, ip, torch.float64))
            sec = rate(ctx, solver, 20, 400)
            algo = A.nnz * (8 + s) + rows * (16 + 2 * s) + 12 * rows * s
            what = f"CG, 2-D 5-pt {nx}^2 fp64 CSR, 1 shard"
        elif name == "bicgstab":
            rows, per, s = 15_625_000, 32, 8
            nnz = rows * per
            g = torch.Generator(device="cuda").manual_seed(20250310)
            row = torch.arange(rows, device="cuda", dtype=torch.int64).repeat_interleave(per)
            col = torch.randint(0, rows, (nnz,), generator=g, device="cuda", dtype=torch.int64)
            val = torch.rand(nnz, generator=g, device="cuda", dtype=torch.float64) * 2 - 1
            col[::per] = torch.arange(rows, device="cuda", dtype=torch.int64)
            val[::per] = 64.0
            space = IndexSpace(0, rows - 1)
            ip = IndexPartition.equal(space, 1)
            A = COOMatrix(ctx, row, col, val, space)
            solver = BiCGStabSolver(planner_for(ctx, A, ip, torch.float64))
            sec = rate(ctx, solver, 3, 30)
            # 2 SpMV (each + zero fill) + 4 dots (one of a vector with itself) + 6 axpy/xpay
            algo = 2 * (nnz * (16 + s) + rows * 3 * s + rows * s) + (3 * 2 + 1) * rows * s + 6 * 3 * rows * s
            what = f"BiCGStab, random row-sorted COO {rows} x {per}/row fp64, 1 shard"
        elif name in ("gmres", "gmres_classical"):
            n1, s, m = 768, 4, 30
            rows = n1 ** 3
            ip = IndexPartition.equal(IndexSpace(0, rows - 1), 1)
            _, offs, cfs = sg.benchmark_stencil(3)
            A = sg.create_linearized_csr_stencil_matrix(ctx, (n1, n1, n1), offs, cfs, ip, torch.float32)
            solver = GMRESSolver(planner_for(ctx, A, ip, torch.float32), m,
                                 "classical" if name == "gmres_classical" else "modified")
            sec = rate(ctx, solver, 1, 3)
            spmv = A.nnz * (8 + s) + rows * (16 + 2 * s)
            dots = m * (m + 1) // 2 * 2 + (m + 1) * 1      # k-j dots read two vectors, norms one
            ew = (m * (m + 1) // 2 + m) * 3 + m * 2 + 3      # axpys, scal, xpay
            algo = (m + 1) * spmv + (dots + ew) * rows * s
            what = f"GMRES({m}) restart cycle, 3-D 7-pt {n1}^3 fp32 CSR, 1 shard" + (
                "; classical Gram-Schmidt option (bytes counted for the reference's sequence)"
                if name == "gmres_classical" else "")
        else:
            raise SystemExit(f"unknown configuration {name}")
        print(json.dumps({"config": name, "what": what, "steps_per_s": 1.0 / sec, "ms_per_step": sec * 1e3,
                          "algorithmic_GB_per_step": algo / 1e9, "GBps": algo / sec / 1e9,
                          "frac_of_8TBps": algo / sec / 8e12}), flush=True)
        del solver, A
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
