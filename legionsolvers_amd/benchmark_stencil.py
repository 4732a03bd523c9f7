"""BenchmarkStencil: the reference's benchmark program (test/BenchmarkStencil.cpp) on the
MI355X path, with its flags:

    -dim {1,2,3,4}   1-D [-1 2 -1] | 2-D 5-point | 3-D 7-point | 3-D 27-point
    -solver {1,2,3}  CG | BiCGStab | GMRES(restart 10)
    -nx -ny -nz      grid size            -vp  number of vector pieces
    -it              iterations           -pt  iterations per trace (kept for flag parity)

Like the reference it solves TWO identical systems that share one matrix in every step
(:201-207), runs 5 warm-up traces, and prints "Achieved X ms per iteration."  Launch one
process per GPU (`python -m torch.distributed.run --nproc-per-node N -m
legionsolvers_amd.benchmark_stencil ...`); -vp must be a multiple of the process count.
"""
from __future__ import annotations

import argparse
import os
import sys
import time

import torch
import torch.distributed as dist

from .context import Context
from .index_space import IndexPartition, IndexSpace
from .partitioned_vector import PartitionedVector
from .solvers import BiCGStabSolver, CGSolver, GMRESSolver
from .square_planner import SquarePlanner
from . import stencil_generator as sg


def grid_extent(dim_flag: int, nx: int, ny: int, nz: int):
    return {1: (nx,), 2: (nx, ny), 3: (nx, ny, nz), 4: (nx, ny, nz)}[dim_flag]


def build_problem(ctx, dim_flag, nx, ny, nz, num_pieces, solver_flag, dtype=torch.float64,
                  num_systems=2, restart=10, matrix=None):
    """Matrix, vectors, planner and solver exactly as top_level_task sets them up
    (test/BenchmarkStencil.cpp:170-211,264,318): rhs = 1, sol = 0.  An already generated
    `matrix` of the same grid and partition is reused instead of being generated again."""
    _, offsets, coeffs = sg.benchmark_stencil(dim_flag)
    extent = grid_extent(dim_flag, nx, ny, nz)
    rows = 1
    for e in extent:
        rows *= e
    partition = IndexPartition.equal(IndexSpace(0, rows - 1), num_pieces)
    if matrix is None:
        matrix = sg.create_linearized_csr_stencil_matrix(ctx, extent, offsets, coeffs, partition, dtype)
    planner = SquarePlanner(ctx, dtype)
    sols, rhss = [], []
    for s in range(num_systems):
        sol = PartitionedVector(ctx, "sol" if s == 0 else f"sol{s}", partition, dtype)
        sol.zero_fill()
        sols.append(sol)
        planner.add_sol_vector(sol)
    for s in range(num_systems):
        rhs = PartitionedVector(ctx, "rhs" if s == 0 else f"rhs{s}", partition, dtype)
        rhs.constant_fill(1.0)
        rhss.append(rhs)
        planner.add_rhs_vector(rhs)
    for s in range(num_systems):
        planner.add_row_partitioned_matrix(matrix, s, s)
    if solver_flag == 1:
        solver = CGSolver(planner)
    elif solver_flag == 2:
        solver = BiCGStabSolver(planner)
    elif solver_flag == 3:
        solver = GMRESSolver(planner, restart)
    else:
        raise ValueError("-solver must be 1 (CG), 2 (BiCGStab) or 3 (GMRES)")
    return matrix, planner, solver


def init_distributed():
    """One process per GPU; rendezvous from the torchrun environment.  LSK_DIST_BACKEND=gloo
    with fewer GPUs than ranks is the rehearsal set-up: ranks share the cards and every
    transfer is staged through host memory (see comm.Communicator)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    backend = os.environ.get("LSK_DIST_BACKEND", "nccl")
    if backend == "gloo":
        local_rank %= max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local_rank)
    if world > 1 and not dist.is_initialized():
        if backend == "nccl":
            try:    # (device_id binds the communicator to this rank's GPU at once instead of at the first collective)
                dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
            except TypeError:   # a torch.distributed without that argument
                dist.init_process_group("nccl")
        else:
            dist.init_process_group(backend)
    return world, local_rank


def timed_steps(ctx, solver, warmup: int, steps: int) -> float:
    """Seconds for `steps` solver steps after `warmup` untimed ones, bracketed by a barrier and
    a device synchronisation on both sides; the maximum over ranks."""
    for _ in range(warmup):
        solver.step()
    ctx.synchronize()
    ctx.comm.barrier()
    ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        solver.step()
    ctx.synchronize()
    ctx.comm.barrier()
    return ctx.comm.max_over_ranks(time.perf_counter() - t0, ctx.device)


def main(argv=None):
    ap = argparse.ArgumentParser(prog="BenchmarkStencil", prefix_chars="-")
    ap.add_argument("-dim", type=int, default=1)
    ap.add_argument("-solver", type=int, default=1)
    ap.add_argument("-nx", type=int, default=16)
    ap.add_argument("-ny", type=int, default=16)
    ap.add_argument("-nz", type=int, default=16)
    ap.add_argument("-vp", type=int, default=4)
    ap.add_argument("-it", type=int, default=20)
    ap.add_argument("-pt", type=int, default=1)
    ap.add_argument("-graph", type=int, default=0,
                    help="1: after the first trace the -pt steps of a trace are recorded once as a HIP graph and every "
                         "further trace is one replay (one rank), as the reference replays its Legion traces")
    ap.add_argument("-systems", type=int, default=2, choices=(1, 2),
                    help="systems stepped per iteration on the one matrix: 2 as the reference's program does "
                         "(test/BenchmarkStencil.cpp:181-199), 1 for a single system (same flag in host_cpp/BenchmarkStencil)")
    args = ap.parse_args(argv)
    world, local_rank = init_distributed()
    ctx = Context(torch.device("cuda", local_rank))
    _, _, solver = build_problem(ctx, args.dim, args.nx, args.ny, args.nz, args.vp, args.solver,
                                 num_systems=args.systems)
    num_warmup_traces = 5
    num_traces = (args.it + args.pt - 1) // args.pt
    warm = num_warmup_traces * args.pt
    timed = max(num_traces - num_warmup_traces, 0) * args.pt
    if args.graph:
        for _ in range(args.pt):     # the first trace as it is
            solver.step()
        recorded = solver.capture(args.pt)
        recorded.replay(num_warmup_traces - 1)
        ctx.synchronize()
        t0 = time.perf_counter()
        recorded.replay(max(num_traces - num_warmup_traces, 0))
        ctx.synchronize()
        seconds = time.perf_counter() - t0
    else:
        seconds = timed_steps(ctx, solver, warm, timed)
    for note in getattr(solver.planner.row_partitioned_matrices[0].matrix, "setup_notes", ()):
        print(f"[LegionSolvers] rank {ctx.rank}: {note[:200]}", file=sys.stderr, flush=True)
    placed = getattr(solver.planner, "workspace_placement", None)
    if placed:
        print(f"[LegionSolvers] rank {ctx.rank}: workspace vectors behind {placed['blocks_in_front']} block(s) of a third of "
              f"the device; product by blocks in front: {placed['product_ms_by_blocks_in_front']} ms", file=sys.stderr, flush=True)
    if ctx.rank == 0:
        print(f"Performed {num_warmup_traces} warmup traces.")
        print(f"Performed {max(num_traces - num_warmup_traces, 0)} timed traces.")
        if timed:
            print(f"Achieved {seconds / timed * 1e3:g}ms per iteration.")
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
