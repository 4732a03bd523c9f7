#!/usr/bin/env python3
"""Kernel A/B timing on one GPU: the CSR SpMV kernel forms (lsk_set_option keys) and the BLAS-1
streams, interleaved rounds in ONE process, HIP events on the launch stream.

    python tools/kbench.py [--n 512] [--rounds 5] [--reps 10] [--dim 4] [--dtype f64]

Prints median / min milliseconds and algorithmic GB/s per configuration.
"""
from __future__ import annotations

import argparse
import itertools
import os
import statistics
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

from legionsolvers_amd import Context, IndexPartition, IndexSpace, PartitionedVector, Scalar  # noqa: E402
from legionsolvers_amd import _lib  # noqa: E402
from legionsolvers_amd import benchmark_stencil as bs  # noqa: E402
from legionsolvers_amd import stencil_generator as sg  # noqa: E402


def timed(fn, reps):
    start, stop = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    start.record()
    for _ in range(reps):
        fn()
    stop.record()
    stop.synchronize()
    return start.elapsed_time(stop) / reps


def coo_bench(args, ctx, lib, dtype, esz):
    from legionsolvers_amd import COOMatrix
    _, offs, cfs = sg.benchmark_stencil(args.dim)
    extent = bs.grid_extent(args.dim, args.n, args.n, args.n)
    rows = 1
    for e in extent:
        rows *= e
    cases = {}
    ip = IndexPartition.equal(IndexSpace(0, rows - 1), 1)
    cases[f"stencil dim-flag {args.dim} n={args.n}"] = (
        sg.create_linearized_coo_stencil_matrix(ctx, extent, offs, cfs, ip, dtype), rows, ip)
    if args.coo_random_rows:
        r = args.coo_random_rows
        g = torch.Generator(device="cuda").manual_seed(20250310)
        row = torch.arange(r, device="cuda", dtype=torch.int64).repeat_interleave(32)
        col = torch.randint(0, r, (r * 32,), generator=g, device="cuda", dtype=torch.int64)
        val = (torch.rand(r * 32, generator=g, device="cuda", dtype=torch.float64) * 2 - 1).to(dtype)
        rip = IndexPartition.equal(IndexSpace(0, r - 1), 1)
        cases[f"random 32/row rows={r}"] = (COOMatrix(ctx, row, col, val, IndexSpace(0, r - 1)), r, rip)
    for name, (A, n, part) in cases.items():
        kp = A.create_kernel_partition_from_range_partition(part)
        x = PartitionedVector(ctx, "x", part, dtype)
        y = PartitionedVector(ctx, "y", part, dtype)
        x.owned.copy_(torch.randn(n, device="cuda", dtype=dtype))
        algo = A.nnz * (16 + esz) + n * 3 * esz
        print(f"{name}: rows={n} nnz={A.nnz} algorithmic bytes/SpMV={algo}", flush=True)

        def spmv():
            A.matvec(y, x.owned, 0, n - 1, kp, 1)

        ref = None
        for v in (0, 1):
            lib.lsk_set_option(b"coo.variant", v)
            y.zero_fill()
            spmv()
            got = y.owned.clone()
            if ref is None:
                ref = got
            else:
                err = float((got - ref).abs().max())
                assert err <= (1e-12 if esz == 8 else 1e-4) * float(ref.abs().max()), err
        for _ in range(50):
            spmv()
        torch.cuda.synchronize()
        times = {0: [], 1: []}
        for _ in range(args.rounds):
            for v in (0, 1):
                lib.lsk_set_option(b"coo.variant", v)
                spmv()
                times[v].append(timed(spmv, args.reps))
        for v in (0, 1):
            med = statistics.median(times[v])
            print(f"  coo.variant={v}: median {med:.3f} ms  min {min(times[v]):.3f} ms  "
                  f"{algo / med / 1e6:.1f} GB/s  ({algo / med / 1e6 / 8000:.3f} of 8 TB/s)", flush=True)
        # the wave-scan form on 32-bit copies of the index fields
        lib.lsk_set_option(b"coo.variant", 1)
        assert A.compress(0, n - 1, kp, part)
        y.zero_fill()
        spmv()
        err = float((y.owned - ref).abs().max())
        assert err <= (1e-12 if esz == 8 else 1e-4) * float(ref.abs().max()), err
        t32 = []
        for _ in range(args.rounds):
            spmv()
            t32.append(timed(spmv, args.reps))
        algo32 = A.nnz * (8 + esz) + n * 3 * esz
        med = statistics.median(t32)
        print(f"  coo.variant=1 on 32-bit index fields: median {med:.3f} ms  min {min(t32):.3f} ms  "
              f"{algo32 / med / 1e6:.1f} GB/s of its own {algo32} B  ({algo32 / med / 1e6 / 8000:.3f} of 8 TB/s)",
              flush=True)
        del A, x, y
        torch.cuda.empty_cache()
    lib.lsk_set_option(b"coo.variant", 1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=512)
    ap.add_argument("--dim", type=int, default=4)
    ap.add_argument("--dtype", default="f64")
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--variants", default="0,1")
    ap.add_argument("--nt", default="0,1")
    ap.add_argument("--remap", default="0,1")
    ap.add_argument("--logrw", default="-1")
    ap.add_argument("--kmax", default="0", help="csr.max_units values (0 = automatic, 4, 8)")
    ap.add_argument("--blas1", action="store_true")
    ap.add_argument("--c32", action="store_true", help="time the compressed-column form of the CSR kernel")
    ap.add_argument("--coo", action="store_true", help="time the COO kernel forms instead (stencil at --n)")
    ap.add_argument("--coo-random-rows", type=int, default=0,
                    help="with --coo: also a random row-sorted matrix, 32 entries per row")
    args = ap.parse_args()
    dtype = torch.float64 if args.dtype == "f64" else torch.float32
    esz = 8 if args.dtype == "f64" else 4
    ctx = Context(torch.device("cuda", 0))
    lib = _lib.lib()
    if args.coo:
        return coo_bench(args, ctx, lib, dtype, esz)
    _, offs, cfs = sg.benchmark_stencil(args.dim)
    extent = bs.grid_extent(args.dim, args.n, args.n, args.n)
    rows = 1
    for e in extent:
        rows *= e
    ip = IndexPartition.equal(IndexSpace(0, rows - 1), 1)
    A = sg.create_linearized_csr_stencil_matrix(ctx, extent, offs, cfs, ip, dtype)
    kp = A.create_kernel_partition_from_range_partition(ip)
    x = PartitionedVector(ctx, "x", ip, dtype)
    y = PartitionedVector(ctx, "y", ip, dtype)
    x.owned.copy_(torch.randn(rows, device="cuda", dtype=dtype))
    algo = A.nnz * (8 + esz) + rows * (16 + 2 * esz)
    print(f"matrix: rows={rows} nnz={A.nnz} algorithmic bytes/SpMV={algo}", flush=True)

    if args.c32:
        assert A.compress(0, rows - 1, kp, ip)
        algo = A.nnz * (4 + esz) + rows * (4 + 2 * esz)
        print(f"compressed columns: algorithmic bytes/SpMV={algo}", flush=True)

    def spmv():
        A.matvec(y, x.owned, 0, rows - 1, kp, 0)

    configs = list(itertools.product([int(v) for v in args.variants.split(",")],
                                     [int(v) for v in args.nt.split(",")],
                                     [int(v) for v in args.remap.split(",")],
                                     [int(v) for v in args.logrw.split(",")],
                                     [int(v) for v in args.kmax.split(",")]))
    configs = [c for c in configs if not (c[0] != 1 and (c[1] == 1 or c[3] != -1 or c[4] != 0))]  # NT / RW / units forms: variant 1 only
    times = {c: [] for c in configs}
    ref = None
    for c in configs:  # results agree across forms
        lib.lsk_set_option(b"csr.variant", c[0])
        lib.lsk_set_option(b"csr.nontemporal", c[1])
        lib.lsk_set_option(b"xcd_remap", c[2])
        lib.lsk_set_option(b"csr.log_rows_per_wave", c[3])
        lib.lsk_set_option(b"csr.max_units", c[4])
        spmv()
        got = y.owned.clone()
        if ref is None:
            ref = got
        else:
            err = float((got - ref).abs().max())
            if True:
                assert err <= 1e-12 * float(ref.abs().max()) * (1 if esz == 8 else 1e6), (c, err)
    for _ in range(150):  # ~2 s of load so that clocks settle before anything is timed
        spmv()
    torch.cuda.synchronize()
    for r in range(args.rounds):
        for c in configs:
            lib.lsk_set_option(b"csr.variant", c[0])
            lib.lsk_set_option(b"csr.nontemporal", c[1])
            lib.lsk_set_option(b"xcd_remap", c[2])
            lib.lsk_set_option(b"csr.log_rows_per_wave", c[3])
            lib.lsk_set_option(b"csr.max_units", c[4])
            spmv()
            times[c].append(timed(spmv, args.reps))
    print(f"{'variant':>8} {'nt':>3} {'remap':>5} {'logrw':>5} {'kmx':>3} {'median ms':>10} {'min ms':>8} {'GB/s(med)':>10} {'frac 8TB/s':>10}")
    for c in configs:
        med, mn = statistics.median(times[c]), min(times[c])
        print("   rounds:", " ".join(f"{t:.2f}" for t in times[c]))
        print(f"{c[0]:>8} {c[1]:>3} {c[2]:>5} {c[3]:>5} {c[4]:>3} {med:>10.3f} {mn:>8.3f} {algo / med / 1e6:>10.1f} {algo / med / 1e6 / 8000:>10.3f}",
              flush=True)

    if args.blas1:
        z = PartitionedVector(ctx, "z", ip, dtype)
        z.owned.copy_(torch.randn(rows, device="cuda", dtype=dtype))
        a = Scalar(ctx, 0.999, dtype)
        ops = {
            "axpy  (3 streams)": (lambda: y.axpy(a, x), 3),
            "xpay  (3 streams)": (lambda: y.xpay(a, x), 3),
            "scal  (2 streams)": (lambda: y.scal(a), 2),
            "dot   (2 streams)": (lambda: x.dot(z), 2),
            "dot vv (1 stream)": (lambda: x.dot(x), 1),
            "copy  (2 streams)": (lambda: y.assign(x), 2),
            "fill  (1 stream) ": (lambda: y.constant_fill(1.0), 1),
            "torch copy_ (2)  ": (lambda: y.owned.copy_(x.owned), 2),
        }
        res = {(k, f): [] for k in ops for f in (0, 1)}
        for r in range(args.rounds):
            for form in (0, 1):
                lib.lsk_set_option(b"blas1.form", form)
                for k, (fn, _) in ops.items():
                    fn()
                    res[(k, form)].append(timed(fn, args.reps))
        for k, (fn, streams) in ops.items():
            for form in (0, 1):
                med = statistics.median(res[(k, form)])
                print(f"{k} form {form}: {med:8.3f} ms  {streams * rows * esz / med / 1e6:8.1f} GB/s", flush=True)


if __name__ == "__main__":
    main()
