#!/usr/bin/env python3
"""Experiment: the first process on a fresh box steps CG at 512^3 at 160.6-161.4 it/s every time (product 4.78-4.80 ms),
later processes on the same box anywhere between 160 and 174.  Is it where the arrays land in HBM?  Modes:
  plain     as it is
  hold N    N GB allocated first and held while the problem is built and stepped
  shift N   N GB allocated first and freed once the problem is built
  twice     the problem built, dropped and built again
  copy N    the matrix as handed over, then N GB (held), then the sliced copy and every vector
  vectors N everything but the solver's workspace vectors, then N GB (held), then p, q and r
Prints the CG rate with the placement the planner chose and with runs of 2 and 64 blocks per XCD forced."""
import gc, os, sys
import torch
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
from legionsolvers_amd import Context  # noqa: E402
from legionsolvers_amd import benchmark_stencil as bs  # noqa: E402


def main():
    mode = sys.argv[1] if len(sys.argv) > 1 else "plain"
    gb = int(sys.argv[2]) if len(sys.argv) > 2 else 64
    ctx = Context(torch.device("cuda", 0))
    dummy = None
    if mode in ("hold", "shift"):
        dummy = torch.empty(gb << 30, dtype=torch.uint8, device="cuda")
    build = lambda **kw: bs.build_problem(ctx, 4, 512, 512, 512, 1, 1, torch.float64, num_systems=1, **kw)  # noqa: E731
    if mode == "copy":      # the matrix as handed over first, then N GB, then the sliced copy and every vector
        from legionsolvers_amd import IndexPartition, IndexSpace
        _, offsets, coeffs = bs.sg.benchmark_stencil(4)
        part = IndexPartition.equal(IndexSpace(0, 512 ** 3 - 1), 1)
        given = bs.sg.create_linearized_csr_stencil_matrix(ctx, (512, 512, 512), offsets, coeffs, part, torch.float64)
        dummy = torch.empty(gb << 30, dtype=torch.uint8, device="cuda")
        matrix, planner, solver = build(matrix=given)
    elif mode == "ranks":   # matrix as handed over in the first 96 GiB, the sliced copy in the second, the workspace vectors in the third
        from legionsolvers_amd import IndexPartition, IndexSpace
        _, offsets, coeffs = bs.sg.benchmark_stencil(4)
        part = IndexPartition.equal(IndexSpace(0, 512 ** 3 - 1), 1)
        given = bs.sg.create_linearized_csr_stencil_matrix(ctx, (512, 512, 512), offsets, coeffs, part, torch.float64)
        torch.cuda.empty_cache()
        total = torch.cuda.mem_get_info()[1]
        used = total - torch.cuda.mem_get_info()[0]
        region = 96 << 30
        held = [torch.empty(max(region - used, 1 << 20), dtype=torch.uint8, device="cuda")]   # fills the first region
        real = bs.CGSolver

        def later(planner):
            torch.cuda.empty_cache()
            used_now = total - torch.cuda.mem_get_info()[0]
            held.append(torch.empty(max(2 * region - used_now, 1 << 20), dtype=torch.uint8, device="cuda"))   # fills the second
            return real(planner)

        bs.CGSolver = later
        matrix, planner, solver = build(matrix=given)
        bs.CGSolver = real
        dummy = held
    elif mode == "vectors":  # everything but the solver's workspace (p, q, r) first, then N GB, then those three
        real = bs.CGSolver
        held = []

        def later(planner):
            held.append(torch.empty(gb << 30, dtype=torch.uint8, device="cuda"))
            return real(planner)

        bs.CGSolver = later
        matrix, planner, solver = build()
        bs.CGSolver = real
        dummy = held
    else:
        matrix, planner, solver = build()
    if mode == "twice":
        del matrix, planner, solver
        gc.collect()
        torch.cuda.empty_cache()
        matrix, planner, solver = build()
    if mode == "shift":
        del dummy
        torch.cuda.empty_cache()
    for _ in range(10):
        solver.step()
    chosen = dict(matrix.sliced_placement)
    out = {}
    for rnd in range(2):
        for place in (chosen[0], 0, 2, 64):
            matrix.sliced_placement[0] = place
            bs.timed_steps(ctx, solver, 3, 0)
            out.setdefault(place, []).append(round(20 / bs.timed_steps(ctx, solver, 0, 20), 1))
    print(f"mode {mode} {gb if mode in ('hold', 'shift', 'copy', 'vectors') else ''}: planner chose {chosen}; CG it/s by placement {out}; "
          f"values at {matrix.sliced[0].handle.value if hasattr(matrix.sliced[0].handle, 'value') else 0:#x}", flush=True)


main()
