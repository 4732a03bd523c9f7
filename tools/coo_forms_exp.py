#!/usr/bin/env python3
"""Experiment: the COO product on the random 32-per-row matrix of 15.6 M rows in row order and in
column-blocked orders of several block sizes (what COOMatrix._try_column_blocks chooses between).

    python tools/coo_forms_exp.py
"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from legionsolvers_amd import Context  # noqa: E402

ctx = Context(torch.device("cuda", 0))
be = ctx.backend
rows, per = 15_625_000, 32
nnz = rows * per
g = torch.Generator(device="cuda").manual_seed(20250310)
row = torch.arange(rows, device="cuda", dtype=torch.int64).repeat_interleave(per)
col = torch.randint(0, rows, (nnz,), generator=g, device="cuda", dtype=torch.int64)
val = torch.rand(nnz, generator=g, device="cuda", dtype=torch.float64) * 2 - 1
col[::per] = torch.arange(rows, device="cuda", dtype=torch.int64)
row32 = be.csr_compress_cols(row, 0, rows - 1)
col32 = be.csr_compress_cols(col, 0, rows - 1)
del row, col
x = torch.rand(rows, dtype=torch.float64, device="cuda")
y = torch.zeros(rows, dtype=torch.float64, device="cuda")


def timed(fn, reps=5):
    fn()
    ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    ctx.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


print(f"row order: {timed(lambda: be.coo_spmv(nnz, row32, col32, val, x, 0, rows - 1, y, 0, rows - 1)):.3f} ms", flush=True)
for block_kib in (512, 1024, 2048, 4096, 8192):
    order = be.coo_block_order(col32, rows, (block_kib << 10) // 8)
    blocked = [be.gather(t, order) for t in (row32, col32, val)]
    del order
    t = timed(lambda: be.coo_spmv(nnz, *blocked, x, 0, rows - 1, y, 0, rows - 1))
    print(f"column blocks of {block_kib} KiB: {t:.3f} ms", flush=True)
    del blocked
    torch.cuda.empty_cache()
