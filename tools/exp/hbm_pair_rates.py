#!/usr/bin/env python3
"""Experiment: does the rate of a two-stream kernel depend on WHICH two parts of HBM it touches?  The device is
allocated in chunks of CHUNK GiB, in order; a copy (read one chunk, write another) is timed for a few source chunks
against every destination chunk.
    python tools/exp/hbm_pair_rates.py [chunk GiB] [source chunk ...]"""
import os, sys
import torch
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
from legionsolvers_amd import Context  # noqa: E402


def timed(fn, reps=6):
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
    sources = [int(a) for a in sys.argv[2:]] or [0, 8, 17, 30]
    ctx = Context(torch.device("cuda", 0))
    be = ctx.backend
    n = (gib << 30) // 8
    chunks = []
    while True:
        free, _ = torch.cuda.mem_get_info()
        if free < (gib << 30) + (1 << 30):
            break
        chunks.append(torch.empty(n, dtype=torch.float64, device="cuda"))
    for c in chunks:
        be.fill(c, 1.0)
    torch.cuda.synchronize()
    print(f"{len(chunks)} chunks of {gib} GiB; copy GB/s (read + write) from the source chunk to every other chunk", flush=True)
    for s in sources:
        if s >= len(chunks):
            continue
        rates = []
        for j, c in enumerate(chunks):
            if j == s:
                rates.append(0.0)
                continue
            rates.append(2 * n * 8 / timed(lambda: be.copy(c, chunks[s])) / 1e6)
        print(f"copy from {s:2d}:", " ".join(f"{r:5.0f}" for r in rates), flush=True)
    out = torch.empty(1, dtype=torch.float64, device="cuda")
    print("dot product GB/s (two read streams) of the source chunk with every other chunk", flush=True)
    for s in sources:
        if s >= len(chunks):
            continue
        rates = [0.0 if j == s else 2 * n * 8 / timed(lambda: be.dot(chunks[s], c, out)) / 1e6 for j, c in enumerate(chunks)]
        print(f"dot with  {s:2d}:", " ".join(f"{r:5.0f}" for r in rates), flush=True)


main()
