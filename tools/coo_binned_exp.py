#!/usr/bin/env python3
"""Experiment: the binned COO product on the random 32-per-row matrix of 15.6 M rows (the BiCGStab
configuration's matrix) for several block / tile sizes, beside the column-blocked scan kernel.

    python tools/coo_binned_exp.py [f64|f32] [blocks:tiles ...]     (default 0:0 = the library's defaults)

Run under `rocprofv3 --kernel-trace --stats` for the time of each pass.
"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from legionsolvers_amd import Context  # noqa: E402

args = sys.argv[1:]
dtype = torch.float32 if args and args[0] == "f32" else torch.float64
if args and args[0] in ("f32", "f64"):
    args = args[1:]
shapes = [tuple(int(v) for v in a.split(":")) for a in args] or [(0, 0)]
esz = 4 if dtype == torch.float32 else 8

ctx = Context(torch.device("cuda", 0))
be = ctx.backend
rows, per = int(os.environ.get("LSK_EXP_ROWS", "15625000")), 32
nnz = rows * per
g = torch.Generator(device="cuda").manual_seed(20250310)
row = torch.arange(rows, device="cuda", dtype=torch.int64).repeat_interleave(per)
col = torch.randint(0, rows, (nnz,), generator=g, device="cuda", dtype=torch.int64)
val = (torch.rand(nnz, generator=g, device="cuda", dtype=torch.float64) * 2 - 1).to(dtype)
col[::per] = torch.arange(rows, device="cuda", dtype=torch.int64)
row32 = be.csr_compress_cols(row, 0, rows - 1)
col32 = be.csr_compress_cols(col, 0, rows - 1)
del row, col
x = torch.rand(rows, dtype=dtype, device="cuda")
y = torch.zeros(rows, dtype=dtype, device="cuda")
c32_bytes = nnz * (8 + esz) + rows * 3 * esz


def timed(fn, reps=5):
    fn()
    ctx.synchronize()
    best = float("inf")
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        ctx.synchronize()
        best = min(best, (time.perf_counter() - t0) / reps * 1e3)
    return best


if os.environ.get("LSK_EXP_SCAN", "1") != "0":
    order = be.coo_block_order(col32, rows, (4 << 20) // esz)
    blocked = [be.gather(t, order) for t in (row32, col32, val)]
    del order
    y.zero_()
    be.coo_spmv(nnz, *blocked, x, 0, rows - 1, y, 0, rows - 1)
    y_scan = y.clone()
    t = timed(lambda: be.coo_spmv(nnz, *blocked, x, 0, rows - 1, y, 0, rows - 1))
    print(f"scan kernel, column blocks of 4 MiB: {t:.3f} ms  {c32_bytes / t / 1e9:.2f} TB/s of {c32_bytes / 1e9:.2f} GB",
          flush=True)
    del blocked
    torch.cuda.empty_cache()
else:
    y_scan = None

n_cal = 1 << 27      # a stream of known size beside the products, to calibrate hardware counters against
cal_a = torch.rand(n_cal, dtype=torch.float64, device="cuda")
cal_b = torch.rand(n_cal, dtype=torch.float64, device="cuda")
cal_out = torch.empty(1, dtype=torch.float64, device="cuda")
for _ in range(3):
    be.dot(cal_a, cal_b, cal_out)
print(f"dot_bytes={2 * n_cal * 8}", flush=True)
del cal_a, cal_b

for bc, tr in shapes:
    t0 = time.perf_counter()
    b = be.coo_binned_create(row32, col32, val, 0, rows - 1, rows, bc, tr)
    ctx.synchronize()
    setup = time.perf_counter() - t0
    if b is None:
        print(f"blocks of {bc}, tiles of {tr}: declined", flush=True)
        continue
    be.coo_binned_spmv(b, x, y, 0)
    diff = float((y - y_scan).abs().max() / y_scan.abs().max()) if y_scan is not None else float("nan")
    t = timed(lambda: be.coo_binned_spmv(b, x, y, 0))
    own = b.own_bytes(rows, rows, esz)
    print(f"binned, blocks of {b.block_cols} x tiles of {b.tile_rows} ({b.blocks} x {b.tiles}, {b.segments} segments, "
          f"{b.units} units, padded {b.padded_entries / nnz - 1:.4%}): {t:.3f} ms  "
          f"{c32_bytes / t / 1e9:.2f} TB/s of the 32-bit layout's {c32_bytes / 1e9:.2f} GB, "
          f"{own / t / 1e9:.2f} TB/s of its own {own / 1e9:.2f} GB; set-up {setup:.2f} s; "
          f"max rel diff vs scan {diff:.2e}", flush=True)
    p1 = b.padded_entries * (2 * esz + 2) + rows * esz + b.units * 16
    p2 = b.padded_entries * (esz + 2) + rows * esz + b.segments * 8
    print(f"binned_bytes pass1={p1} pass2={p2} c32_layout={c32_bytes} rows={rows} nnz={nnz} padded={b.padded_entries}",
          flush=True)
    b.close()
    del b
    torch.cuda.empty_cache()
