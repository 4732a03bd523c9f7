#!/usr/bin/env python3
"""The sliced product's time by workgroup placement (runs of c consecutive blocks per XCD) on a benchmark stencil:
    python tools/exp/product_by_placement.py <dim flag> <nx> <ny> <nz> <f32|f64> [c ...]"""
import os, sys
import torch
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
from legionsolvers_amd import Context  # noqa: E402
from legionsolvers_amd import benchmark_stencil as bs  # noqa: E402


def main():
    dim, nx, ny, nz = (int(a) for a in sys.argv[1:5])
    dtype = torch.float32 if sys.argv[5] == "f32" else torch.float64
    cands = [int(a) for a in sys.argv[6:]] or [0, 2, 4, 8, 9, 16, 24, 64, 72]
    ctx = Context(torch.device("cuda", 0))
    matrix, planner, solver = bs.build_problem(ctx, dim, nx, ny, nz, 1, 1, dtype, num_systems=1)
    be = ctx.backend
    s = matrix.sliced[0]
    rows = matrix.rows
    x = planner.get_vector(2, 0)._own
    y = planner.get_vector(3, 0)._own
    print(f"rows {rows} blocks {s.blocks} tuner chose {dict(matrix.sliced_placement)} algorithmic GB {s.algorithmic_bytes(rows, x.element_size()) / 1e9:.3f}", flush=True)
    for rnd in range(3):
        line = []
        for c in cands:
            for _ in range(2):
                be.csr_sliced_spmv(s, 0, rows, x, y, 0, c)
            t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            t0.record()
            for _ in range(10):
                be.csr_sliced_spmv(s, 0, rows, x, y, 0, c)
            t1.record()
            t1.synchronize()
            line.append(f"{c}: {t0.elapsed_time(t1) / 10:.3f}")
        print("  ".join(line), flush=True)


main()
