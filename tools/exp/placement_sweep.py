#!/usr/bin/env python3
"""Launch time of the sliced product (one vector, and two where served) against the workgroup -> XCD
placement: runs of N consecutive blocks per XCD (0 = blocks dealt round robin).

    python tools/exp/placement_sweep.py [n] [dim-flag] [f32|f64] [placements...]
"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
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
    flag = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    dtype = torch.float32 if (len(sys.argv) > 3 and sys.argv[3] == "f32") else torch.float64
    places = [int(a) for a in sys.argv[4:]] or [0, 4, 16, 64, 256, 1024]
    os.environ["LSK_TUNE"] = "0"
    ctx = Context(torch.device("cuda", 0))
    matrix, planner, _ = bs.build_problem(ctx, flag, n, n, n, 1, 1, dtype, num_systems=1)
    planner.get_vector(2, 0).random_fill(1)
    s = matrix.sliced[0]
    be = ctx.backend
    xs = [planner.get_vector(2, 0)._own, planner.get_vector(2, 0)._own.clone()]
    ys = [planner.get_vector(3, 0)._own, planner.get_vector(3, 0)._own.clone()]
    out = {"grid": n, "dim_flag": flag, "blocks": s.blocks, "block_rows": s.block_rows}
    for rep in range(2):
        for p in places:
            out[f"one/{p}/{rep}"] = round(timed(ctx, lambda: be.csr_sliced_spmv(s, 0, matrix.rows, xs[0], ys[0], 0, p), 10), 3)
            if s.pair_served:
                out[f"two/{p}/{rep}"] = round(timed(ctx, lambda: be.csr_sliced_spmv2(s, matrix.rows, xs, ys, 0, placement=p), 10), 3)
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
