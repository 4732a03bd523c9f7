#!/usr/bin/env python3
"""Experiment: the fused BLAS-1 leaves of a CG step on 1 GiB fp64 vectors, by which 96 GB part of HBM each operand
lies in.  Vectors a0..a2 are allocated first (part A on a fresh device), then a block of a third of the device, then
b0..b2 (part B).  Whether the two groups really differ is checked by a copy between them (faster across parts)."""
import os, sys
import torch
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
from legionsolvers_amd import Context  # noqa: E402


def timed(fn, reps=20):
    for _ in range(3):
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
    be = ctx.backend
    n = 1 << 27
    a = [torch.zeros(n, dtype=torch.float64, device="cuda") for _ in range(3)]
    third = torch.cuda.mem_get_info()[1] // 3
    block = torch.empty(third, dtype=torch.uint8, device="cuda")
    b = [torch.zeros(n, dtype=torch.float64, device="cuda") for _ in range(3)]
    alpha = be.new_scalar(1e-3, torch.float64)
    out = torch.zeros(1, dtype=torch.float64, device="cuda")
    partials = torch.zeros(1 << 16, dtype=torch.float64, device="cuda")
    print(f"copy inside A {timed(lambda: be.copy(a[1], a[0])):.4f} ms, A -> B {timed(lambda: be.copy(b[0], a[0])):.4f} ms, "
          f"inside B {timed(lambda: be.copy(b[1], b[0])):.4f} ms", flush=True)
    cases = {
        "dot(v, w)": lambda v, w, z: be.dot(v, w, out),
        "axpy y += a x (y = first, x = second)": lambda y, x, z: be.axpy(y, [alpha], x),
        "axpy_sqnorm y += a x, |y|^2 (y = first, x = second)": lambda y, x, z: be.axpy_sqnorm(y, [alpha], x, partials),
        "xpay y = x + a y (y = first, x = second)": lambda y, x, z: be.xpay(y, [alpha], x),
    }
    for name, fn in cases.items():
        same = timed(lambda: fn(b[0], b[1], None))
        across = timed(lambda: fn(b[0], a[1], None))
        print(f"{name}: both in B {same:.4f} ms, second operand in A {across:.4f} ms ({(across / same - 1) * 100:+.1f} %)", flush=True)
    # axpy_xpay(z, fa, p, fb, r): z += a p; p = r + b p
    combos = {"z p r all in B": (b[0], b[1], b[2]), "z in A, p r in B": (a[0], b[1], b[2]),
              "z r in A, p in B": (a[0], b[1], a[2]), "z p in B, r in A": (b[0], b[1], a[2])}
    for name, (z, p, r) in combos.items():
        print(f"axpy_xpay {name}: {timed(lambda: be.axpy_xpay(z, [alpha], p, [alpha], r)):.4f} ms", flush=True)
    del block


main()
