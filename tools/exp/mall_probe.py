#!/usr/bin/env python3
"""Experiment: what the Infinity Cache does for a buffer that is written and read again soon.  For sizes
from 16 MB to 2 GB: fill (write only), dot v.v (read only), copy, each repeated on the same buffers, with
plain and with non-temporal accesses.  Rates above what HBM gives (~5 TB/s writes, ~7 TB/s reads) are the cache's.

    python tools/exp/mall_probe.py
"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from legionsolvers_amd import Context, _lib  # noqa: E402
from legionsolvers_amd.backend import _ptr  # noqa: E402

ctx = Context(torch.device("cuda", 0))
be = ctx.backend
out = torch.empty(1, dtype=torch.float64, device="cuda")


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    ctx.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


for mb in (16, 32, 64, 96, 128, 192, 256, 512, 2048):
    n = mb * (1 << 20) // 8
    a = torch.zeros(n, dtype=torch.float64, device="cuda")
    b = torch.zeros(n, dtype=torch.float64, device="cuda")
    row = [f"{mb:5d} MB"]
    for nt in (0, 1):
        _lib.call("lsk_set_option", b"blas1.nontemporal", nt)
        t_fill = timed(lambda: _lib.call("lsk_fill_f64", n, 1.5, _ptr(a), be.stream()))
        t_dot = timed(lambda: be.dot(a, a, out))
        t_copy = timed(lambda: _lib.call("lsk_copy_f64", n, _ptr(a), _ptr(b), be.stream()))
        # write then read the same buffer, alternating: what a producer / consumer pair of kernels sees
        def pair():
            _lib.call("lsk_fill_f64", n, 1.5, _ptr(a), be.stream())
            be.dot(a, a, out)
        t_pair = timed(pair)
        gb = n * 8 / 1e9
        row.append(f"{'nt' if nt else 'plain'}: fill {gb / t_fill:6.2f} TB/s  read {gb / t_dot:6.2f}  copy {2 * gb / t_copy:6.2f}  "
                   f"fill+read {2 * gb / t_pair:6.2f}")
    print(" | ".join(row), flush=True)
    del a, b
_lib.call("lsk_set_option", b"blas1.nontemporal", -1)
