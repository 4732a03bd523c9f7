#!/usr/bin/env python3
"""Driver of tools/exp/blas1_exp.hip: the BLAS-1 streams with plain / nontemporal loads and stores on
vectors of n elements (default 2^27 = 1 GiB of fp64, the headline problem's vectors).

    python tools/blas1_exp.py --build
    python tools/blas1_exp.py [--n 134217728] [--reps 20]
"""
import argparse
import ctypes
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "exp", "blas1_exp.hip")
LIB = os.path.join(HERE, "exp", "libblas1_exp.so")
OPS = {0: ("axpy", 3), 1: ("dot", 2), 2: ("axpy+xpay", 5), 3: ("copy", 2), 4: ("dot(v,v)", 1)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--build", action="store_true")
    ap.add_argument("--n", type=int, default=1 << 27)
    ap.add_argument("--reps", type=int, default=20)
    args = ap.parse_args()
    if args.build:
        subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off",
                        "-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops", SRC, "-o", LIB], check=True)
        return
    import torch
    lib = ctypes.CDLL(LIB)
    P = lambda t: ctypes.c_void_p(t.data_ptr())
    stream = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    n = args.n
    x, y, z = (torch.rand(n, dtype=torch.float64, device="cuda") for _ in range(3))
    out = torch.zeros(8192, dtype=torch.float64, device="cuda")
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for op, (name, passes) in OPS.items():
        for ntl, nts in ((0, 0), (1, 0), (0, 1), (1, 1)):
            if nts and op in (1, 4):
                continue
            for _ in range(3):
                assert lib.blas1_exp_launch(op, ntl, nts, n, P(x), P(y), P(z), P(out), stream()) == 0
            best = 1e9
            for _ in range(3):
                t0.record()
                for _ in range(args.reps):
                    lib.blas1_exp_launch(op, ntl, nts, n, P(x), P(y), P(z), P(out), stream())
                t1.record()
                t1.synchronize()
                best = min(best, t0.elapsed_time(t1) / args.reps)
            print(f"{name:10s} loads {'nt   ' if ntl else 'plain'} stores {'nt   ' if nts else 'plain'}  {best:.4f} ms  "
                  f"{passes * n * 8 / best / 1e9:.2f} TB/s", flush=True)


if __name__ == "__main__":
    main()
