#!/usr/bin/env python3
"""CG steps per second on the 27-point stencil at several grid sizes with the BLAS-1 streams' non-temporal /
one-chunk-per-workgroup form forced on (blas1.nontemporal = 1), forced off (0) and by the library's size rule (-1):
where is the vector size from which the streaming form wins?"""
import json, os, sys, time
import torch
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
from legionsolvers_amd import Context, _lib  # noqa: E402
from legionsolvers_amd import benchmark_stencil as bs  # noqa: E402

def main():
    ctx = Context(torch.device("cuda", 0))
    sizes = [int(a) for a in sys.argv[1:]] or [64, 96, 128, 160, 192, 224, 256, 320]
    for n1 in sizes:
        _, _, solver = bs.build_problem(ctx, 4, n1, n1, n1, 1, 1, torch.float64, num_systems=1)
        out = {"grid": n1, "vector_MiB": round(n1 ** 3 * 8 / 2 ** 20, 1)}
        steps = max(20, min(400, int(3e9 / (n1 ** 3 * 300))))
        for opt in (0, 1, -1, 0, 1):
            _lib.call("lsk_set_option", b"blas1.nontemporal", opt)
            for _ in range(5):
                solver.step()
            ctx.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                solver.step()
            ctx.synchronize()
            rate = steps / (time.perf_counter() - t0)
            out.setdefault(f"nt={opt}", []).append(round(rate, 1))
        _lib.call("lsk_set_option", b"blas1.nontemporal", -1)
        print(json.dumps(out), flush=True)
        del solver
        torch.cuda.empty_cache()

main()
