#!/usr/bin/env python3
"""Solver steps launched one by one against the same steps replayed from a recorded trace (HIP graph),
on the benchmark configurations of one GPU: python tools/graph_rates.py [cg2d] [cg3d] [gmres] [bicgstab_stencil]
Prints one JSON line per configuration."""
import json
import os
import sys
import time

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from legionsolvers_amd import Context  # noqa: E402
from legionsolvers_amd import benchmark_stencil as bs  # noqa: E402

CONFIGS = {
    # name: (dim flag, nx, ny, nz, solver flag, dtype, restart, steps)
    "cg_small": (4, 64, 64, 64, 1, torch.float64, 10, 400),
    "cg2d": (2, 4096, 4096, 1, 1, torch.float64, 10, 200),
    "cg3d": (4, 512, 512, 512, 1, torch.float64, 10, 20),
    "bicgstab_stencil": (4, 256, 256, 256, 2, torch.float64, 10, 50),
    "gmres": (3, 768, 768, 768, 3, torch.float32, 30, 2),
}


def rate(ctx, run, steps):
    ctx.synchronize()
    t0 = time.perf_counter()
    run(steps)
    ctx.synchronize()
    return steps / (time.perf_counter() - t0)


def main():
    names = sys.argv[1:] or ["cg_small", "cg2d", "cg3d"]
    ctx = Context(torch.device("cuda", 0))
    for name in names:
        dim, nx, ny, nz, solver_flag, dtype, restart, steps = CONFIGS[name]
        _, _, solver = bs.build_problem(ctx, dim, nx, ny, nz, 1, solver_flag, dtype, num_systems=1, restart=restart)
        for _ in range(3):
            solver.step()
        eager = rate(ctx, lambda k: [solver.step() for _ in range(k)], steps)
        recorded = solver.capture()
        recorded.replay(2)
        replayed = rate(ctx, recorded.replay, steps)
        eager_again = rate(ctx, lambda k: [solver.step() for _ in range(k)], steps)
        print(json.dumps({"config": name, "grid": [nx, ny, nz], "solver": solver_flag, "dtype": str(dtype).split(".")[-1],
                          "steps": steps, "steps_per_s_eager": eager, "steps_per_s_replayed": replayed,
                          "steps_per_s_eager_afterwards": eager_again, "gain": replayed / max(eager, eager_again)}),
              flush=True)
        del recorded, solver
        torch.cuda.empty_cache()


main()
