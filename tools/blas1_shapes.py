#!/usr/bin/env python3
"""Driver of tools/exp/blas1_shapes.hip: launch shapes of the BLAS-1 streams on vectors of n fp64 elements.

    python tools/blas1_shapes.py --build
    python tools/blas1_shapes.py [--n 134217728] [--reps 20]
"""
import argparse
import ctypes
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "exp", "blas1_shapes.hip")
LIB = os.path.join(HERE, "exp", "libblas1_shapes.so")
OPS = {0: ("axpy", 3), 5: ("axpy+sqnorm", 3), 2: ("axpy+xpay", 5), 3: ("copy", 2)}


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
    lib.blas1_shapes_launch.argtypes = [ctypes.c_int] * 4 + [ctypes.c_int64] * 2 + [ctypes.c_void_p] * 5
    P = lambda t: ctypes.c_void_p(t.data_ptr())
    stream = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    n = args.n
    x, y, z = (torch.rand(n, dtype=torch.float64, device="cuda") for _ in range(3))
    out = torch.zeros(4096, dtype=torch.float64, device="cuda")
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    shapes = []
    for u, block in ((4, 256), (8, 256), (2, 256), (4, 512), (4, 1024), (2, 1024), (8, 128), (16, 64)):
        for slab in (0, 1):
            for blocks in (1024, 2048, 4096, 16384, 0):
                shapes.append((u, block, slab, blocks))
    for op, (name, passes) in OPS.items():
        results = []
        for u, block, slab, blocks in shapes:
            nchunks = n // 2 // (u * block)
            grid = nchunks if blocks == 0 else min(blocks, nchunks)
            if slab and blocks == 0:
                continue
            args_ = (op, u, block, slab, grid, n, P(x), P(y), P(z), P(out), stream())
            rc = lib.blas1_shapes_launch(*args_)
            if rc != 0:
                continue
            for _ in range(2):
                lib.blas1_shapes_launch(*args_)
            best = 1e9
            for _ in range(3):
                t0.record()
                for _ in range(args.reps):
                    lib.blas1_shapes_launch(*args_)
                t1.record()
                t1.synchronize()
                best = min(best, t0.elapsed_time(t1) / args.reps)
            results.append((best, u, block, slab, grid))
        results.sort()
        print(f"== {name} ({passes} streams of {n * 8 / 2**20:.0f} MiB)", flush=True)
        for best, u, block, slab, grid in results[:6] + results[-2:]:
            print(f"  {best:.4f} ms  {passes * n * 8 / best / 1e9:.2f} TB/s   {u} x 16 B per lane, {block} lanes, "
                  f"{'slab per workgroup' if slab else 'round-robin chunks'}, {grid} workgroups", flush=True)
        ref = [r for r in results if r[1:4] == (4, 256, 0) and r[4] == 2048]
        if ref:
            print(f"  (the library's shape, 4 x 16 B, 256 lanes, round robin, 2048 workgroups: {ref[0][0]:.4f} ms)", flush=True)


if __name__ == "__main__":
    main()
