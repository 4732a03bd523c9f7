#!/usr/bin/env python3
"""Can one CG step of the Python planner be captured in a HIP graph and replayed?  Small grid, one rank:
warm-up steps eagerly, one step captured, replays compared with an eager solver that took as many steps."""
import os, sys, time
import torch
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
from legionsolvers_amd import Context  # noqa: E402
from legionsolvers_amd import benchmark_stencil as bs  # noqa: E402

def main():
    n1 = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    ctx = Context(torch.device("cuda", 0))
    _, _, eager = bs.build_problem(ctx, 4, n1, n1, n1, 1, 1, torch.float64, num_systems=1)
    _, _, solver = bs.build_problem(ctx, 4, n1, n1, n1, 1, 1, torch.float64, num_systems=1)
    warm, replays = 4, 6
    for _ in range(warm):
        solver.step()
    for _ in range(warm + 1 + replays):
        eager.step()
    want = [r.get_value() for r in eager.residual_norm_squared]
    ctx.synchronize()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        with torch.cuda.graph(g, stream=side):
            solver.step()
    torch.cuda.current_stream().wait_stream(side)
    print("captured; history entries", len(solver.residual_norm_squared), flush=True)
    got = []
    for i in range(1 + replays):   # (capture does not execute: the first replay IS step warm + 1)
        g.replay()
        torch.cuda.synchronize()
        got.append(solver.residual_norm_squared[-1].get_value())
        _ = torch.zeros(1000, device="cuda").sum().item()   # eager work between the replays
    print("want", want[warm + 1:], flush=True)
    print("got ", got, flush=True)
    print("equal", got == want[warm + 1:], flush=True)
    # host cost per step: eager against replay
    t0 = time.perf_counter()
    for _ in range(200):
        eager.step()
    torch.cuda.synchronize()
    t_eager = (time.perf_counter() - t0) / 200
    t0 = time.perf_counter()
    for _ in range(200):
        g.replay()
    torch.cuda.synchronize()
    t_graph = (time.perf_counter() - t0) / 200
    print(f"n={n1}^3: eager {t_eager * 1e6:.1f} us per step, graph replay {t_graph * 1e6:.1f} us per step", flush=True)

main()
