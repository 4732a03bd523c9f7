#!/usr/bin/env python3
"""Experiment: is every part of HBM equally fast?  The whole device is allocated in chunks of CHUNK GiB, in order, and a
read stream (the library's dot product of a chunk with itself: non-temporal 16-byte loads, one pass) and a write stream
(fill) are timed on each chunk alone.
    python tools/exp/hbm_region_rates.py [chunk GiB]"""
import os, sys
import torch
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
from legionsolvers_amd import Context  # noqa: E402


def timed(fn, reps=8):
    fn()
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(reps):
        fn()
    t1.record()
    t1.synchronize()
    return t0.elapsed_time(t1) / reps


def main():
    gib = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    ctx = Context(torch.device("cuda", 0))
    be = ctx.backend
    n = (gib << 30) // 8
    chunks = []
    while True:
        free, _ = torch.cuda.mem_get_info()
        if free < (gib << 30) + (1 << 30):
            break
        chunks.append(torch.empty(n, dtype=torch.float64, device="cuda"))
    out = torch.empty(1, dtype=torch.float64, device="cuda")
    for c in chunks:
        be.fill(c, 1.0)
    torch.cuda.synchronize()
    print(f"{len(chunks)} chunks of {gib} GiB", flush=True)
    for rnd in range(2):
        reads, writes = [], []
        for c in chunks:
            reads.append(n * 8 / timed(lambda: be.dot(c, c, out)) / 1e6)
            writes.append(n * 8 / timed(lambda: be.fill(c, 1.0)) / 1e6)
        print("read  GB/s by chunk:", " ".join(f"{r:5.0f}" for r in reads), flush=True)
        print("write GB/s by chunk:", " ".join(f"{w:5.0f}" for w in writes), flush=True)


main()
