#!/usr/bin/env python3
"""Experiment: would the in-place updates (y += a x: two read streams and a write stream into one of them) run faster
out of place, writing into another 96 GB part of HBM than they read from?  torch's own element-wise kernel on 1 GiB
fp64 vectors, so only the ratios count: in place; into a third vector in the same part; into one in another part."""
import torch


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
    n = 1 << 27
    a = [torch.zeros(n, dtype=torch.float64, device="cuda") for _ in range(3)]
    third = torch.cuda.mem_get_info()[1] // 3
    block = torch.empty(third, dtype=torch.uint8, device="cuda")
    b = [torch.zeros(n, dtype=torch.float64, device="cuda") for _ in range(3)]
    print(f"copy inside the first part {timed(lambda: a[1].copy_(a[0])):.4f} ms, across {timed(lambda: b[0].copy_(a[0])):.4f} ms", flush=True)
    print(f"y += a x in place, both in one part: {timed(lambda: b[0].add_(b[1], alpha=0.5)):.4f} ms", flush=True)
    print(f"z = y + a x, all three in one part: {timed(lambda: torch.add(b[0], b[1], alpha=0.5, out=b[2])):.4f} ms", flush=True)
    print(f"z = y + a x, z in the other part: {timed(lambda: torch.add(b[0], b[1], alpha=0.5, out=a[2])):.4f} ms", flush=True)
    print(f"z = y + a x, y and z in one part, x in the other: {timed(lambda: torch.add(b[0], a[1], alpha=0.5, out=b[2])):.4f} ms", flush=True)
    del block


main()
