#!/usr/bin/env python3
"""Experiment: the two-vector sliced product (27-point 512^3 fp64) by where its four vectors lie: in front (the part
of HBM the matrix and its copy fill on a fresh device) or behind a block of a third of the device.  The pair's
workgroup placement is taken over the candidates for each layout, as the tuner would."""
import os, sys
import torch
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
os.environ["LSK_PLACE_VECTORS"] = "0"
from legionsolvers_amd import Context  # noqa: E402
from legionsolvers_amd import benchmark_stencil as bs  # noqa: E402


def timed(fn, reps=5):
    fn()
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(reps):
        fn()
    t1.record()
    t1.synchronize()
    return t0.elapsed_time(t1) / reps


def main():
    ctx = Context(torch.device("cuda", 0))
    matrix, planner, solver = bs.build_problem(ctx, 4, 512, 512, 512, 1, 1, torch.float64, num_systems=1)
    be = ctx.backend
    rows = matrix.rows
    handle = matrix.sliced[0]
    g = torch.Generator(device="cuda").manual_seed(1)
    front = [torch.randn(rows, dtype=torch.float64, device="cuda", generator=g) for _ in range(4)]
    third = torch.cuda.mem_get_info()[1] // 3
    block = torch.empty(third, dtype=torch.uint8, device="cuda")
    back = [torch.randn(rows, dtype=torch.float64, device="cuda", generator=g) for _ in range(4)]
    F, B = front, back
    layouts = {
        "xa xb ya yb in front": (F[0], F[1], F[2], F[3]),
        "xa xb ya yb behind": (B[0], B[1], B[2], B[3]),
        "x in front, y behind": (F[0], F[1], B[2], B[3]),
        "x behind, y in front": (B[0], B[1], F[2], F[3]),
        "system a in front, system b behind": (F[0], B[1], F[2], B[3]),
    }
    single_f = timed(lambda: be.csr_sliced_spmv(handle, 0, rows, F[0], F[2], 0, matrix.sliced_placement.get(0, 0)))
    single_b = timed(lambda: be.csr_sliced_spmv(handle, 0, rows, B[0], B[2], 0, matrix.sliced_placement.get(0, 0)))
    print(f"single product: in front {single_f:.3f} ms, behind {single_b:.3f} ms", flush=True)
    for name, (xa, xb, ya, yb) in layouts.items():
        by_place = {}
        for place in (0, 2, 4, 16, 64):
            by_place[place] = round(timed(lambda: be.csr_sliced_spmv2(handle, rows, (xa, xb), (ya, yb), 0, placement=place)), 3)
        best = min(by_place, key=by_place.get)
        print(f"pair, {name}: best {by_place[best]:.3f} ms (runs of {best}); {by_place}", flush=True)
    slots = be.csr_sliced_dot_slots(handle, rows)
    parts = (torch.zeros(slots, dtype=torch.float64, device="cuda"), torch.zeros(slots, dtype=torch.float64, device="cuda"))
    for name in ("xa xb ya yb in front", "xa xb ya yb behind"):
        xa, xb, ya, yb = layouts[name]
        by_place = {}
        for place in (0, 2, 4, 16, 64):
            by_place[place] = round(timed(lambda: be.csr_sliced_spmv2(handle, rows, (xa, xb), (ya, yb), 0, dot_with=(xa, xb),
                                                                      partials=parts, placement=place)), 3)
        best = min(by_place, key=by_place.get)
        print(f"pair with the shares of ya.xa and yb.xb, {name}: best {by_place[best]:.3f} ms (runs of {best}); {by_place}", flush=True)
    del block


main()
