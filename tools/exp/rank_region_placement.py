#!/usr/bin/env python3
"""Experiment: CG at 512^3 with the sliced copy's values and the solver's vectors put into chosen thirds of HBM.  The
three thirds (96 GiB each on a 288 GiB device; presumably the three ranks of the 12-high stacks) are told apart by
probing: a copy between two buffers of one third runs at 5.6-5.9 TB/s, between two thirds at 6.2 TB/s
(tools/exp/hbm_pair_rates.py).  Method: after the matrix as handed over is generated, ALL free memory is taken in
chunks of 8 GiB and every chunk classified; before each allocation of the set-up only chunks of the wanted third
are released (and what is left of the hole is filled again afterwards), so the allocation has nowhere else to go.
    python tools/exp/rank_region_placement.py <third of sol/rhs> <third of the copy> <third of p, q, r>     (0, 1, 2)"""
import gc, os, sys
import torch
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
from legionsolvers_amd import Context, IndexPartition, IndexSpace  # noqa: E402
from legionsolvers_amd import benchmark_stencil as bs  # noqa: E402

GIB = 1 << 30


def timed(fn, reps=4):
    fn()
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(reps):
        fn()
    t1.record()
    t1.synchronize()
    return t0.elapsed_time(t1) / reps


def main():
    want_x, want_copy, want_work = (int(a) for a in sys.argv[1:4])
    ctx = Context(torch.device("cuda", 0))
    be = ctx.backend
    _, offsets, coeffs = bs.sg.benchmark_stencil(4)
    part = IndexPartition.equal(IndexSpace(0, 512 ** 3 - 1), 1)
    given = bs.sg.create_linearized_csr_stencil_matrix(ctx, (512, 512, 512), offsets, coeffs, part, torch.float64)
    gc.collect()
    torch.cuda.empty_cache()
    n = 8 * GIB // 8
    chunks = []
    while torch.cuda.mem_get_info()[0] >= 9 * GIB:
        chunks.append(torch.empty(n, dtype=torch.float64, device="cuda"))
    for c in chunks:
        be.fill(c, 1.0)

    def same_third(a, b):   # (whole chunks: a copy of 1 GiB does not tell)
        rate = 2 * n * 8 / timed(lambda: be.copy(a, b)) / 1e6
        return rate < 6050.0, rate

    # classify: thirds are numbered in the order their first chunk appears
    refs, third = [], []
    for c in chunks:
        for k, ref in enumerate(refs):
            if ref is c or same_third(c, ref)[0]:
                third.append(k)
                break
        else:
            refs.append(c)
            third.append(len(refs) - 1)
    print(f"{len(chunks)} chunks of 8 GiB after the matrix; thirds by chunk: {''.join(str(t) for t in third)}", flush=True)
    if len(refs) != 3:
        print("did not find three thirds", flush=True)
    pool = {k: [c for c, t in zip(chunks, third) if t == k] for k in range(len(refs))}
    del chunks, refs
    fillers = []

    def open_hole(k, gib):
        torch.cuda.synchronize()
        for _ in range((gib + 7) // 8):
            pool[k].pop()
        gc.collect()
        torch.cuda.empty_cache()

    def close_holes():
        torch.cuda.synchronize()
        while torch.cuda.mem_get_info()[0] >= (1 << 29):
            free = torch.cuda.mem_get_info()[0]
            try:
                fillers.append(torch.empty(free - (1 << 28), dtype=torch.uint8, device="cuda"))
            except torch.cuda.OutOfMemoryError:
                try:
                    fillers.append(torch.empty(free // 2, dtype=torch.uint8, device="cuda"))
                except torch.cuda.OutOfMemoryError:
                    break

    real_slice = type(given).slice_structure
    real_solver = bs.CGSolver

    def slice_later(self, *args, **kw):
        close_holes()
        open_hole(want_copy, 32)
        return real_slice(self, *args, **kw)

    def solver_later(planner):
        close_holes()
        open_hole(want_work, 8)
        return real_solver(planner)

    type(given).slice_structure = slice_later
    bs.CGSolver = solver_later
    open_hole(want_x, 8)
    os.environ["LSK_TUNE"] = "0"      # (no temporary vectors at set-up; the placement is forced below)
    matrix, planner, solver = bs.build_problem(ctx, 4, 512, 512, 512, 1, 1, torch.float64, num_systems=1, matrix=given)
    type(given).slice_structure = real_slice
    bs.CGSolver = real_solver
    for _ in range(10):
        solver.step()
    out = {}
    for rnd in range(2):
        for place in (0, 2, 64):
            matrix.sliced_placement[0] = place
            bs.timed_steps(ctx, solver, 3, 0)
            out.setdefault(place, []).append(round(20 / bs.timed_steps(ctx, solver, 0, 20), 1))
    print(f"sol/rhs in third {want_x}, the copy in third {want_copy}, p q r in third {want_work}: CG it/s by placement {out}"
          f"{'' if matrix.sliced else '  (NO sliced copy: ' + '; '.join(n[:80] for n in matrix.setup_notes) + ')'}", flush=True)


main()
