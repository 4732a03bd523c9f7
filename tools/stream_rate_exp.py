#!/usr/bin/env python3
"""Experiment: what a plain read stream achieves on a 29-GB array (the size of the sliced-pattern
copy's values) against a 2-GB one: lsk_dot_f64(v, v) reads n * 8 bytes.

    python tools/stream_rate_exp.py
"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from legionsolvers_amd import Context  # noqa: E402

ctx = Context(torch.device("cuda", 0))
be = ctx.backend
out = be.empty(1, torch.float64)
for n in (1 << 28, 3_614_447_616):
    v = torch.zeros(n, dtype=torch.float64, device="cuda")
    for rep in range(3):
        be.dot(v, v, out)
        ctx.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            be.dot(v, v, out)
        ctx.synchronize()
        ms = (time.perf_counter() - t0) / 10 * 1e3
        print(f"n = {n}: {ms:.3f} ms per pass, {n * 8 / ms / 1e9:.3f} TB/s", flush=True)
    del v
    torch.cuda.empty_cache()
    time.sleep(2)
