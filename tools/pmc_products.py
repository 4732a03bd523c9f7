#!/usr/bin/env python3
"""Workload for rocprofv3 --pmc passes over the OTHER products the benchmark configurations run (the headline's
three layouts are tools/pmc_spmv.py's): the two-vector sliced product (27-point 512^3 fp64, two systems), the
7-point 768^3 fp32 product of the GMRES configuration and the 2-D 5-point 4096^2 fp64 product, each launched by a
few solver steps, plus dot products of known size to calibrate FETCH_SIZE against.

    rocprofv3 --pmc FETCH_SIZE --output-format csv -d out -- python3 tools/pmc_products.py
    rocprofv3 --pmc WRITE_SIZE --output-format csv -d out -- python3 tools/pmc_products.py
    python3 tools/pmc_products_traffic.py <fetch csv> <write csv> <this script's output> <out.json>

Prints, per product, the kernel-name prefix to look for and the algorithmic bytes of one launch."""
import gc
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from legionsolvers_amd import Context  # noqa: E402
from legionsolvers_amd import benchmark_stencil as bs  # noqa: E402

ctx = Context(torch.device("cuda", 0))


def run(label, dim, shape, solver_flag, dtype, systems, steps, kernel, vectors_per_launch, **kw):
    matrix, planner, solver = bs.build_problem(ctx, dim, shape[0], shape[1], shape[2], 1, solver_flag, dtype,
                                               num_systems=systems, **kw)
    for _ in range(steps):
        solver.step()
    ctx.synchronize()
    s = matrix.sliced[0]
    esz = 4 if dtype == torch.float32 else 8
    rows = matrix.rows
    # one launch: the padded values once, and per vector its x once and its y once
    algo = s.algorithmic_bytes(rows, esz) + (vectors_per_launch - 1) * 2 * rows * esz
    place = matrix.pair_placement if systems == 2 else matrix.sliced_placement
    print(f"product {label} | {kernel} | {algo} | placement {dict(place)}", flush=True)
    del matrix, planner, solver, s
    gc.collect()
    torch.cuda.empty_cache()


run("pair_27pt_512_f64", 4, (512, 512, 512), 1, torch.float64, 2, 6, "sliced_wave_kernel<double, 27, true, 2,", 2)
run("cg2d_5pt_4096_f64", 2, (4096, 4096, 1), 1, torch.float64, 1, 6, "sliced_wave_kernel<double, 8, true, 1,", 1)
run("gmres_7pt_768_f32", 3, (768, 768, 768), 3, torch.float32, 1, 1, "sliced_wave_kernel<float, 8, false, 1,", 1, restart=6)
n = 1 << 27
v, w = torch.randn(n, dtype=torch.float64, device="cuda"), torch.randn(n, dtype=torch.float64, device="cuda")
out = torch.empty(1, dtype=torch.float64, device="cuda")
for _ in range(3):
    ctx.backend.dot(v, w, out)
ctx.synchronize()
print(f"calibration dot_kernel<double | {2 * n * 8}", flush=True)
