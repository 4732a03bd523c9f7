#!/usr/bin/env python3
"""The sliced product's workgroup -> XCD placement with SHORT runs of blocks per XCD (2, 4, 8) beside the
candidates the tuner times (0 = round robin, 16, 64, 256, 1024, 4096): ms per product, 27-point 512^3 fp64."""
import json, os, sys
import torch
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
from legionsolvers_amd import Context  # noqa: E402
from legionsolvers_amd import benchmark_stencil as bs  # noqa: E402


def timed(fn, reps):
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fn(); fn()
    t0.record()
    for _ in range(reps):
        fn()
    t1.record()
    t1.synchronize()
    return t0.elapsed_time(t1) / reps


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    os.environ["LSK_TUNE"] = "0"
    ctx = Context(torch.device("cuda", 0))
    matrix, planner, solver = bs.build_problem(ctx, 4, n, n, n, 1, 1, torch.float64, num_systems=1)
    be, s = ctx.backend, matrix.sliced[0]
    rows = matrix.rows
    x = torch.randn(rows, device="cuda", dtype=torch.float64)
    y = torch.empty(rows, device="cuda", dtype=torch.float64)
    shares = torch.empty(be.csr_sliced_dot_slots(s, rows), dtype=torch.float64, device="cuda")
    out = {"grid": n}
    w = torch.randn(rows, device="cuda", dtype=torch.float64)
    for rnd in range(3):
        for place in (0, 2, 3, 4, 8, 16):
            ms = timed(lambda: be.csr_sliced_spmv_dot(s, rows, x, y, 0, x, shares, place), 10)
            out.setdefault(f"dot_w_is_x_place{place}", []).append(round(ms, 3))
            ms = timed(lambda: be.csr_sliced_spmv(s, 0, rows, x, y, 0, place), 10)
            out.setdefault(f"plain_place{place}", []).append(round(ms, 3))
            ms = timed(lambda: be.csr_sliced_spmv_dot(s, rows, x, y, 0, w, shares, place), 10)
            out.setdefault(f"dot_other_w_place{place}", []).append(round(ms, 3))
    for k, v in out.items():
        print(k, v, flush=True)


main()
