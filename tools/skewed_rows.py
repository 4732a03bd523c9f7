#!/usr/bin/env python3
"""CSR products on a matrix with a power-law row-length distribution (a few rows tens of thousands of
times the mean): lsk_csr_spmv_c32_* (the 32-bit structure the planner multiplies general CSR matrices
through) and lsk_csr_spmv_* (the reference layout as handed over), checked against torch.segment_reduce, timed with HIP events.

    python tools/skewed_rows.py [--rows 10000000] [--mean 32] [--alpha 1.5] [--cap 1000000]
                                [--columns local|random] [--dtype f64]

Row lengths: floor(L0 u^(-1/alpha)) for uniform u, capped; L0 chosen for the mean.  Columns "local":
a row's entries are consecutive columns around its own index (x is read like a band matrix's, so the
figure shows what the row lengths cost, not what random gathers cost); "random": uniform columns.
Prints one JSON line: rows, entries, longest rows, milliseconds, GB/s on each layout's own bytes
(c32: nnz (4 + s) + rows (4 + 2 s); int64: nnz (8 + s) + rows (16 + 2 s)) and the fraction of 8 TB/s."""
import argparse
import ctypes
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from legionsolvers_amd import Context, _lib  # noqa: E402


def P(t):
    return ctypes.c_void_p(t.data_ptr())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=10_000_000)
    ap.add_argument("--mean", type=float, default=32.0)
    ap.add_argument("--alpha", type=float, default=1.5)
    ap.add_argument("--cap", type=int, default=1_000_000)
    ap.add_argument("--columns", default="local")
    ap.add_argument("--dtype", default="f64")
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--threshold", type=int, default=0, help="rows longer than this go to the split product's chunk kernels (0: the library's default)")
    ap.add_argument("--log-rows", type=int, default=-1, help="option csr.log_rows_per_wave (rows per wave = 2^k; -1: from the mean row length)")
    args = ap.parse_args()
    if args.log_rows >= 0:
        _lib.call("lsk_set_option", b"csr.log_rows_per_wave", args.log_rows)
    def note(what):
        torch.cuda.synchronize()
        print(f"[skewed_rows] {what}", file=sys.stderr, flush=True)

    ctx = Context(torch.device("cuda", 0))
    dtype = torch.float64 if args.dtype == "f64" else torch.float32
    s, suf = (8, "f64") if dtype == torch.float64 else (4, "f32")
    rows = args.rows
    g = torch.Generator(device="cuda").manual_seed(20251114)
    u = torch.rand(rows, generator=g, device="cuda", dtype=torch.float64).clamp_min(1e-12)
    l0 = args.mean * (args.alpha - 1.0) / args.alpha           # mean of the uncapped law is l0 alpha / (alpha - 1)
    lens = torch.clamp((l0 * u.pow(-1.0 / args.alpha)).floor().to(torch.int64), max=args.cap)
    del u
    offs = torch.zeros(rows + 1, dtype=torch.int64, device="cuda")
    torch.cumsum(lens, 0, out=offs[1:])
    nnz = int(offs[-1])
    top = torch.topk(lens, 5).values.tolist()
    note(f"{rows} rows, {nnz} entries, longest {top}")
    row_of = torch.repeat_interleave(torch.arange(rows, device="cuda", dtype=torch.int64), lens)
    if args.columns == "local":
        within = torch.arange(nnz, device="cuda", dtype=torch.int64) - offs[:-1][row_of]
        col = (row_of - lens[row_of] // 2 + within) % rows
        del within
    else:
        col = torch.randint(0, rows, (nnz,), generator=g, device="cuda", dtype=torch.int64)
    val = (torch.rand(nnz, generator=g, device="cuda", dtype=torch.float64) * 2 - 1).to(dtype)
    x = torch.randn(rows, generator=g, device="cuda", dtype=torch.float64).to(dtype)
    prod = val.double() * x.double()[col]
    want = torch.segment_reduce(prod, "sum", lengths=lens, unsafe=True)      # (sequential per row, fp64)
    bound = torch.segment_reduce(prod.abs(), "sum", lengths=lens, unsafe=True)
    del prod
    del row_of
    note("reference product made")
    rects = torch.stack([offs[:-1], offs[1:] - 1], dim=1).contiguous()
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    col32 = torch.empty(nnz, dtype=torch.int32, device="cuda")
    off32 = torch.empty(rows + 1, dtype=torch.int32, device="cuda")
    _lib.call("lsk_csr_compress_cols_u32", nnz, P(col), 0, rows - 1, P(col32), stream)
    _lib.call("lsk_csr_compress_rows_u32", rows, nnz, P(rects), 0, P(off32), stream)
    y = torch.empty(rows, dtype=dtype, device="cuda")

    def c32():
        _lib.call(f"lsk_csr_spmv_c32_{suf}", rows, nnz, P(off32), P(col32), P(val), P(x), rows, P(y), 0, stream)

    plan = ctypes.c_void_p(0)
    code = _lib.lib().lsk_csr_long_rows_create(rows, P(off32), args.threshold, stream, ctypes.byref(plan))
    info = (ctypes.c_int64 * 4)()
    if code == _lib.LSK_OK:
        _lib.call("lsk_csr_long_rows_info", plan, info)

    def split():
        _lib.call(f"lsk_csr_spmv_c32_split_{suf}", plan, 0, rows, nnz, P(off32), P(col32), P(val), P(x), rows, P(y), 0, stream)

    def i64():
        _lib.call(f"lsk_csr_spmv_{suf}", rows, nnz, P(rects), 0, P(col), P(val), P(x), 0, rows - 1, P(y), 0, stream)

    out = {"rows": rows, "entries": nnz, "mean_row": nnz / rows, "longest_rows": top, "columns": args.columns,
           "dtype": args.dtype, "threshold": int(info[2]) if code == _lib.LSK_OK else None, "log_rows_per_wave": args.log_rows}
    tol = 1e-13 if s == 8 else 2e-6
    out["rows_listed_for_the_split_product"], out["their_chunks"] = int(info[0]), int(info[1])
    forms = [("c32", c32, nnz * (4 + s) + rows * (4 + 2 * s)), ("int64", i64, nnz * (8 + s) + rows * (16 + 2 * s))]
    if code == _lib.LSK_OK:   # (+ 12 bytes of plan and a chunk sum written and read per chunk)
        forms.insert(0, ("c32_split", split, nnz * (4 + s) + rows * (4 + 2 * s) + int(info[1]) * (4 + 2 * s) + int(info[0]) * 12))
    for name, fn, nbytes in forms:
        y.fill_(7.0)
        fn()
        note(f"{name}: first product done")
        err = (y.double() - want).abs()
        assert bool((err <= tol * (bound + 1e-300) * 4).all()), (name, float((err / (bound + 1e-300)).max()))
        for _ in range(3):
            fn()
        t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0.record()
        for _ in range(args.reps):
            fn()
        t1.record()
        t1.synchronize()
        ms = t0.elapsed_time(t1) / args.reps
        out[name] = {"ms": ms, "bytes": nbytes, "GBps": nbytes / ms / 1e6, "frac_of_8TBps": nbytes / ms / 1e6 / 8000}
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
