#!/usr/bin/env python3
"""Solver step rates and per-leaf rooflines of the reference's other benchmark configurations on ONE
GPU (the headline configuration is bench.py's):

  cg2d      Test06CSRSolveCG: CG, 2-D 5-point Laplacian 4096^2, fp64 CSR, one shard
  bicgstab  BiCGStab on the random nonsymmetric row-sorted COO matrix (15 625 000 rows x 32), fp64
  gmres     GMRES(30) on the 3-D 7-point stencil 768^3, fp32 CSR (a step = one restart cycle:
            31 SpMV, 496 dot, 465 axpy, 30 scal, 1 xpay, 30 axpy into the solution)

    python tools/config_rates.py [cg2d] [bicgstab] [gmres] [gmres_classical]

(gmres_classical: the same cycle with GMRESSolver's classical Gram-Schmidt option, every inner product
of a sweep in one pass over w: different arithmetic, the reference's sequence is `gmres`.)

Prints one JSON line per configuration:

  steps_per_s, ms_per_step      plain timed steps (no instrumentation)
  executed                      the bytes of the leaf calls the step ACTUALLY made, each priced on the
                                layout it read (a sliced-pattern copy, a binned copy, fused leaves: fewer
                                passes than the reference's sequence), over the step time: a fraction of
                                the HBM peak that is <= 1 by construction
  reference_sequence            the bytes the reference's own sequence of leaf tasks would move on its
                                own layout (int64 indices, one task per operation), and how many of them
                                the step did not move ("GB_avoided_per_step") -- never priced as a rate
  leaves                        per kind of leaf call, from a second, instrumented run of the same steps
                                (HIP events around every call on the launch stream): launches per step,
                                bytes and mean milliseconds per launch, GB/s and fraction of peak; sorted
                                by their share of the step
"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from legionsolvers_amd import (BiCGStabSolver, CGSolver, COOMatrix, Context, GMRESSolver,  # noqa: E402
                               IndexPartition, IndexSpace, PartitionedVector, SquarePlanner)
from legionsolvers_amd import stencil_generator as sg  # noqa: E402

PEAK = 8e12


class Tally:
    """A HipBackend seen through a counter: every leaf call goes through unchanged; the calls that move
    vectors or matrices are tallied with the bytes they must move on the layout they read and, when
    `timed` is set, bracketed by HIP events on the launch stream."""

    def __init__(self, backend):
        object.__setattr__(self, "_be", backend)
        object.__setattr__(self, "calls", {})    # name -> [launches, bytes, [(start, stop) events]]
        object.__setattr__(self, "timed", False)
        object.__setattr__(self, "on", False)

    def __getattr__(self, name):
        attr = getattr(self._be, name)
        price = getattr(self, "_bytes_" + name, None)
        if price is None or not callable(attr):
            return attr

        def call(*args, **kwargs):
            if not self.on:
                return attr(*args, **kwargs)
            label, nbytes = price(*args, **kwargs)
            entry = self.calls.setdefault(label, [0, 0, []])
            if self.timed:
                start, stop = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                start.record()
                out = attr(*args, **kwargs)
                stop.record()
                entry[2].append((start, stop))
            else:
                out = attr(*args, **kwargs)
            entry[0] += 1
            entry[1] += nbytes
            return out

        return call

    def __setattr__(self, name, value):
        if name in ("timed", "on", "calls"):
            object.__setattr__(self, name, value)
        else:
            setattr(self._be, name, value)

    def reset(self):
        self.calls = {}

    # -- what each leaf must move (s = element size) ---------------------------------------
    @staticmethod
    def _s(t):
        return t.element_size()

    def _bytes_fill(self, x, value):
        return "fill", x.numel() * self._s(x)

    def _bytes_copy(self, dst, src):
        return "copy", 2 * dst.numel() * self._s(dst)

    def _bytes_scal(self, x, futs):
        return "scal", 2 * x.numel() * self._s(x)

    def _bytes_axpy(self, y, futs, x):
        return "axpy", 3 * y.numel() * self._s(y)

    def _bytes_xpay(self, y, futs, x):
        return "xpay", 3 * y.numel() * self._s(y)

    def _bytes_axpy_sqnorm(self, y, futs, x, partials):
        return "axpy + squared norm", 3 * y.numel() * self._s(y)

    def _bytes_axpy_dot(self, y, futs, x, w, partials):
        if w is None:
            return "axpy + squared norm", 3 * y.numel() * self._s(y)
        return "axpy + dot", 4 * y.numel() * self._s(y)

    def _bytes_axpy_xpay(self, z, futs_a, p, futs_b, r):
        return "axpy + xpay", 5 * p.numel() * self._s(p)

    def _bytes_dot(self, v, w, out):
        same = v.data_ptr() == w.data_ptr()
        return ("dot (v.v)" if same else "dot"), (1 if same else 2) * v.numel() * self._s(v)

    def _bytes_multi_dot(self, vs, w, out):
        k = len(vs)
        return f"multi_dot ({k} vectors)", (k + -(-k // 8)) * w.numel() * self._s(w)

    def _bytes_multi_axpy(self, y, coef, xs, negate, partials=None):
        k = len(xs)
        return f"multi_axpy ({k} vectors)", (k + 2 * -(-k // 8)) * y.numel() * self._s(y)

    def _bytes_csr_spmv(self, rows, nnz, rects, k_base, col, val, x, x_lo, x_hi, y, beta, placement=0):
        s = self._s(val)
        if col.dtype == torch.int32:
            return "CSR product, 32-bit structure", nnz * (4 + s) + rows * (4 + (2 + beta) * s)
        return "CSR product, int64 structure", nnz * (8 + s) + rows * (16 + (2 + beta) * s)

    def _bytes_csr_spmv_dot(self, rows, nnz, rects, k_base, col, val, x, x_lo, x_hi, y, beta, w, partials,
                            placement=0):
        label, nbytes = self._bytes_csr_spmv(rows, nnz, rects, k_base, col, val, x, x_lo, x_hi, y, beta)
        return label + " + dot", nbytes + rows * self._s(val)

    @staticmethod
    def _sliced_bytes(sliced, rows, s):
        piece_rows = sliced.blocks * sliced.block_rows
        whole = sliced.algorithmic_bytes(min(piece_rows, max(rows, 0)), s)
        return whole if rows >= piece_rows - sliced.block_rows else int(whole * rows / max(piece_rows, 1))

    def _bytes_csr_sliced_spmv(self, sliced, row_begin, row_end, x, y, beta, placement=0):
        s, rows = self._s(y), row_end - row_begin
        return "CSR product, sliced-pattern copy", self._sliced_bytes(sliced, rows, s) + beta * rows * s

    def _bytes_csr_sliced_spmv_dot(self, sliced, row_end, x, y, beta, w, partials, placement=0, row_begin=0):
        s, rows = self._s(y), row_end - row_begin
        # (the shares' second operand is taken from the staged windows of x when it is x itself)
        own_w = 0 if (w.data_ptr() >= x.data_ptr() and w.data_ptr() < x.data_ptr() + x.numel() * s) else rows * s
        return "CSR product, sliced-pattern copy + dot", self._sliced_bytes(sliced, rows, s) + beta * rows * s + own_w

    def _bytes_csr_sliced_spmv2(self, sliced, rows, xs, ys, beta, dot_with=None, partials=None, placement=0,
                                row_begin=0):
        s, n = self._s(ys[0]), rows - row_begin
        one = self._sliced_bytes(sliced, n, s)
        return "two CSR products in one pass, sliced-pattern copy", one + 2 * n * s + 2 * beta * n * s

    def _bytes_coo_spmv(self, nnz, row, col, val, x, x_lo, x_hi, y, y_lo, y_hi):
        s, rows = self._s(val), y_hi - y_lo + 1
        index = 8 if row.dtype == torch.int32 else 16
        return f"COO product, scan kernel, {index // 2 * 8}-bit indices", nnz * (index + s) + rows * 3 * s

    def _bytes_coo_binned_spmv(self, binned, x, y, beta):
        s = self._s(y)
        return "COO product, binned copy (both passes)", binned.own_bytes(y.numel(), x.numel(), s) + beta * y.numel() * s


def planner_for(ctx, matrix, ip, dtype):
    rhs = PartitionedVector(ctx, "rhs", ip, dtype)
    rhs.constant_fill(1.0)
    sol = PartitionedVector(ctx, "sol", ip, dtype)
    sol.zero_fill()
    pl = SquarePlanner(ctx, dtype)
    pl.add_sol_vector(sol)
    pl.add_rhs_vector(rhs)
    pl.add_row_partitioned_matrix(matrix, 0, 0)
    return pl


def rate(ctx, solver, warm, steps):
    for _ in range(warm):
        solver.step()
    ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        solver.step()
    ctx.synchronize()
    return (time.perf_counter() - t0) / steps


def leaves_of(ctx, solver, steps):
    """The instrumented run: `steps` more steps with every leaf call tallied and timed."""
    tally = ctx.backend
    tally.reset()
    tally.on = tally.timed = True
    for _ in range(steps):
        solver.step()
    ctx.synchronize()
    tally.on = tally.timed = False
    out, total_bytes = [], 0
    for label, (launches, nbytes, events) in tally.calls.items():
        ms = sum(a.elapsed_time(b) for a, b in events) / max(len(events), 1)
        total_bytes += nbytes
        out.append({"leaf": label, "launches_per_step": launches / steps, "GB_per_launch": nbytes / launches / 1e9,
                    "ms_per_launch": ms, "ms_per_step": ms * launches / steps,
                    "GBps": nbytes / launches / (ms * 1e-3) / 1e9 if ms > 0 else None,
                    "frac_of_8TBps": nbytes / launches / (ms * 1e-3) / PEAK if ms > 0 else None})
    out.sort(key=lambda e: -e["ms_per_step"])
    return out, total_bytes / steps


def main():
    which = sys.argv[1:] or ["cg2d", "bicgstab", "gmres"]
    ctx = Context(torch.device("cuda", 0))
    ctx.backend = Tally(ctx.backend)
    for name in which:
        if name == "cg2d":
            nx, s = 4096, 8
            rows = nx * nx
            ip = IndexPartition.equal(IndexSpace(0, rows - 1), 1)
            _, offs, cfs = sg.benchmark_stencil(2)
            A = sg.create_linearized_csr_stencil_matrix(ctx, (nx, nx), offs, cfs, ip)
            solver = CGSolver(planner_for(ctx, A, ip, torch.float64))
            sec = rate(ctx, solver, 20, 400)
            tally_steps = 50
            # SpMV + fill + 2 dots + 2 axpy + xpay, one task each (src/CGSolver.hpp:46-55)
            ref = A.nnz * (8 + s) + rows * (16 + 2 * s) + 14 * rows * s
            what = f"CG, 2-D 5-pt {nx}^2 fp64 CSR, 1 shard"
        elif name == "bicgstab":
            rows, per, s = 15_625_000, 32, 8
            nnz = rows * per
            g = torch.Generator(device="cuda").manual_seed(20250310)
            row = torch.arange(rows, device="cuda", dtype=torch.int64).repeat_interleave(per)
            col = torch.randint(0, rows, (nnz,), generator=g, device="cuda", dtype=torch.int64)
            val = torch.rand(nnz, generator=g, device="cuda", dtype=torch.float64) * 2 - 1
            col[::per] = torch.arange(rows, device="cuda", dtype=torch.int64)
            val[::per] = 64.0
            space = IndexSpace(0, rows - 1)
            ip = IndexPartition.equal(space, 1)
            A = COOMatrix(ctx, row, col, val, space)
            solver = BiCGStabSolver(planner_for(ctx, A, ip, torch.float64))
            sec = rate(ctx, solver, 3, 30)
            tally_steps = 10
            # 2 SpMV (each + zero fill) + 4 dots (one of a vector with itself) + 6 axpy/xpay
            # (src/BiCGStabSolver.hpp:36-82)
            ref = 2 * (nnz * (16 + s) + rows * 3 * s + rows * s) + (3 * 2 + 1) * rows * s + 6 * 3 * rows * s
            what = f"BiCGStab, random row-sorted COO {rows} x {per}/row fp64, 1 shard"
        elif name in ("gmres", "gmres_classical"):
            n1, s, m = 768, 4, 30
            rows = n1 ** 3
            ip = IndexPartition.equal(IndexSpace(0, rows - 1), 1)
            _, offs, cfs = sg.benchmark_stencil(3)
            A = sg.create_linearized_csr_stencil_matrix(ctx, (n1, n1, n1), offs, cfs, ip, torch.float32)
            solver = GMRESSolver(planner_for(ctx, A, ip, torch.float32), m,
                                 "classical" if name == "gmres_classical" else "modified")
            sec = rate(ctx, solver, 1, 3)
            tally_steps = 2
            spmv = A.nnz * (8 + s) + rows * (16 + 2 * s) + rows * s   # + the zero fill in front of it
            dots = m * (m + 1) // 2 * 2 + (m + 1) * 1      # k-j dots read two vectors, norms one
            ew = (m * (m + 1) // 2 + m) * 3 + m * 2 + 3      # axpys, scal, xpay (src/GMRESSolver.hpp:83-127)
            ref = (m + 1) * spmv + (dots + ew) * rows * s
            what = f"GMRES({m}) restart cycle, 3-D 7-pt {n1}^3 fp32 CSR, 1 shard" + (
                "; classical Gram-Schmidt option (reference_sequence: the reference's modified sweep)"
                if name == "gmres_classical" else "")
        else:
            raise SystemExit(f"unknown configuration {name}")
        leaves, executed = leaves_of(ctx, solver, tally_steps)
        print(json.dumps({
            "config": name, "what": what, "steps_per_s": 1.0 / sec, "ms_per_step": sec * 1e3,
            "executed": {"GB_per_step": executed / 1e9, "GBps": executed / sec / 1e9,
                         "frac_of_8TBps": executed / sec / PEAK},
            "reference_sequence": {"GB_per_step": ref / 1e9, "GB_avoided_per_step": (ref - executed) / 1e9},
            "leaves": leaves}), flush=True)
        del solver, A
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
