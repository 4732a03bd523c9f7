#!/usr/bin/env python3
"""Experiment on the real kernels: does the binned COO product gain from being cut into row slabs whose
intermediate (the products pass 1 writes and pass 2 reads back, 8 bytes per entry) stays in the 256 MB
Infinity Cache?  The random 32-per-row matrix of 15.6 M rows is given K binned copies, copy k holding the
entries of the k-th slab of rows (the row window of lsk_coo_binned_create does the cut); one product is
then the K copies' products one after the other -- slab k's pass 1, slab k's pass 2, slab k + 1's pass 1,
... -- which is the slab-alternating schedule.  What it costs is visible in the same run: every slab's
pass 1 stages all blocks of x again.

    python tools/exp/coo_slabs_exp.py [K ...]        (default 1 4 10 20 40 80)

Prints, per K, the milliseconds of one whole product, the intermediate's size per slab, and the result's
largest difference from K = 1."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from legionsolvers_amd import Context  # noqa: E402

# LSK_EXP_TILES=256: every slab in 256 tiles of rows (one round of pass-2 workgroups over the CUs) instead of
# the library's choice (the largest tiles that fit the LDS, which leaves a thin slab with a few dozen)
tiles_per_slab = int(os.environ.get("LSK_EXP_TILES", "0"))
ks = [int(a) for a in sys.argv[1:]] or [1, 4, 10, 20, 40, 80]
ctx = Context(torch.device("cuda", 0))
be = ctx.backend
rows, per = int(os.environ.get("LSK_EXP_ROWS", "15625000")), 32
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
reference = None
for k in ks:
    bounds = [rows * i // k for i in range(k + 1)]
    copies = []
    for i in range(k):
        # (row-sorted entries: slab i's entries are the run [32 bounds[i], 32 bounds[i + 1]))
        a, b = per * bounds[i], per * bounds[i + 1]
        slab_rows = bounds[i + 1] - bounds[i]
        tile_rows = min(-(-(-(-slab_rows // tiles_per_slab)) // 8) * 8, 20472) if tiles_per_slab else 0
        copies.append((be.coo_binned_create(row32[a:b], col32[a:b], val[a:b], bounds[i], bounds[i + 1] - 1, rows,
                                            0, tile_rows), bounds[i], bounds[i + 1]))
    assert all(c is not None for c, _, _ in copies)

    def product():
        for c, lo, hi in copies:
            be.coo_binned_spmv(c, x, y[lo:hi], 0)

    product()
    ctx.synchronize()
    if reference is None:
        reference = y.clone()
    diff = float((y - reference).abs().max())
    best = float("inf")
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(5):
            product()
        ctx.synchronize()
        best = min(best, (time.perf_counter() - t0) / 5 * 1e3)
    units = sum(c.units for c, _, _ in copies)
    print(f"K = {k:3d} slabs: {best:.3f} ms per product; intermediate {8 * nnz / k / 1e6:.0f} MB per slab; "
          f"{units} workgroups of pass 1 in all (each stages {copies[0][0].block_cols * 8 // 1024} KiB of x), "
          f"{sum(c.tiles for c, _, _ in copies)} of pass 2 (tiles of {copies[0][0].tile_rows} rows); "
          f"largest difference from K = {ks[0]}: {diff:.2e}", flush=True)
    for c, _, _ in copies:
        c.close()
    del copies
    torch.cuda.empty_cache()
