#!/usr/bin/env python3
"""The two-vector sliced product's launch time against the workgroup -> XCD placement (runs of N consecutive
blocks per XCD; 0 = blocks dealt round robin), 27-point 512^3 fp64, twice over.

    python tools/exp/pair_placement_sweep.py
"""
import os, sys, time, json, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from legionsolvers_amd import Context, _lib
from legionsolvers_amd import benchmark_stencil as bs
def timed(ctx, fn, reps):
    fn(); ctx.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    ctx.synchronize(); return (time.perf_counter() - t0) / reps * 1e3
ctx = Context(torch.device("cuda", 0))
matrix, planner, solver = bs.build_problem(ctx, 4, 512, 512, 512, 1, 1, torch.float64, num_systems=1)
planner.get_vector(2, 0).random_fill(1)
s = matrix.sliced[0]; be = ctx.backend
xs = [planner.get_vector(2, 0)._own, planner.get_vector(2, 0)._own.clone()]
ys = [planner.get_vector(3, 0)._own, planner.get_vector(3, 0)._own.clone()]
out = {}
for rep in range(2):
    for p in (0, 4, 8, 16, 32, 64, 128, 256, 512, 1024):
        out[f"{p}/{rep}"] = round(timed(ctx, lambda: be.csr_sliced_spmv2(s, matrix.rows, xs, ys, 0, placement=p), 10), 3)
print(json.dumps(out))
