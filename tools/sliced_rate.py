#!/usr/bin/env python3
"""Launch time of the sliced-pattern CSR product beside the 32-bit-structure product on the
headline matrix (27-point stencil, fp64, one GPU).

    LSK_TUNE_VERBOSE=1 python tools/sliced_rate.py [n] [dim-flag] [f32|f64]
    (grid n^3, default 512; BenchmarkStencil's -dim flag, default 4 = 27-point, 3 = 7-point)
"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from legionsolvers_amd import Context, _lib  # noqa: E402
from legionsolvers_amd import benchmark_stencil as bs  # noqa: E402


def timed(ctx, fn, reps):
    fn()
    ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    ctx.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    flag = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    dtype = torch.float32 if (len(sys.argv) > 3 and sys.argv[3] == "f32") else torch.float64
    esz = 4 if dtype == torch.float32 else 8
    ctx = Context(torch.device("cuda", 0))
    if "LSK_RUNS" in os.environ:
        _lib.call("lsk_set_option", b"sliced.runs", int(os.environ["LSK_RUNS"]))
    t0 = time.perf_counter()
    matrix, planner, solver = bs.build_problem(ctx, flag, n, n, n, 1, 1, dtype, num_systems=1)
    ctx.synchronize()
    setup = time.perf_counter() - t0
    planner.get_vector(2, 0).random_fill(1)
    out = {"grid": n, "dim_flag": flag, "entry_bytes": esz, "rows": matrix.rows, "nnz": matrix.nnz,
           "setup_s": round(setup, 2)}
    if matrix.sliced:
        s = matrix.sliced[0]
        out["sliced_ms"] = round(timed(ctx, lambda: planner.matvec(3, 2), 20), 3)
        out["sliced_bytes"] = s.algorithmic_bytes(matrix.rows, esz)
        out["sliced_TBps"] = round(out["sliced_bytes"] / out["sliced_ms"] / 1e9, 3)
        out.update(patterns=s.patterns, mixed_blocks=s.mixed_blocks, masked_waves=s.masked_waves, blocks=s.blocks, padded_entries=s.padded_entries,
                   placement=matrix.sliced_placement)
        # two right-hand sides with one pass over the values (lsk_csr_sliced_spmv2_*), both forms
        be = ctx.backend
        xs = [planner.get_vector(2, 0)._own, planner.get_vector(2, 0)._own.clone()]
        ys = [planner.get_vector(3, 0)._own, planner.get_vector(3, 0)._own.clone()]
        for pair_form in (1, 0, 2):   # 2: wave-staged with the windows straight into LDS
            _lib.call("lsk_set_option", b"sliced.pair_form", min(pair_form, 1))
            _lib.call("lsk_set_option", b"sliced.dma", 1 if pair_form == 2 else 0)
            if be.csr_sliced_spmv2(s, matrix.rows, xs, ys, 0, placement=matrix.sliced_placement.get(0, 0)) is not None:
                for place2 in (0, 64):   # workgroup -> XCD placements
                    ms = timed(ctx, lambda: be.csr_sliced_spmv2(s, matrix.rows, xs, ys, 0, placement=place2), 10)
                    out[f"pair_form{pair_form}_place{place2}_ms"] = round(ms, 3)
                    if pair_form == 2:   # both right-hand sides the same vector: what the second one's cache footprint costs
                        ms = timed(ctx, lambda: be.csr_sliced_spmv2(s, matrix.rows, [xs[0], xs[0]], ys, 0, placement=place2), 10)
                        out[f"pair_form2_same_x_place{place2}_ms"] = round(ms, 3)
        _lib.call("lsk_set_option", b"sliced.pair_form", -1)
        _lib.call("lsk_set_option", b"sliced.dma", 1)
        # both forms of the one-vector product, plain and with the fused dot shares, called directly
        place = matrix.sliced_placement.get(0, 0)
        shares = torch.empty(be.csr_sliced_dot_slots(s, matrix.rows), dtype=dtype, device="cuda")
        for form, label in ((0, "block_form"), (1, "wave_form"), (1, "wave_dma_form")):
            _lib.call("lsk_set_option", b"sliced.form", form)
            _lib.call("lsk_set_option", b"sliced.dma", 1 if "dma" in label else 0)
            for place1 in (64,):
                out[f"{label}_place{place1}_ms"] = round(timed(ctx, lambda: be.csr_sliced_spmv(s, 0, matrix.rows, xs[0], ys[0], 0, place1), 10), 3)
            out[f"{label}_ms"] = round(timed(ctx, lambda: be.csr_sliced_spmv(s, 0, matrix.rows, xs[0], ys[0], 0, place), 20), 3)
            out[f"{label}_dot_ms"] = round(timed(ctx, lambda: be.csr_sliced_spmv_dot(s, matrix.rows, xs[0], ys[0], 0, xs[1],
                                                                                   shares, place), 20), 3)
            # ... with w the product's own input vector (CG's p . A p): taken from the staged windows, not read again
            out[f"{label}_dot_w_is_x_ms"] = round(timed(ctx, lambda: be.csr_sliced_spmv_dot(s, matrix.rows, xs[0], ys[0], 0, xs[0],
                                                                                          shares, place), 20), 3)
        _lib.call("lsk_set_option", b"sliced.dma", 1)
        _lib.call("lsk_set_option", b"sliced.form", int(os.environ.get("LSK_SLICED_FORM", "1")))
        # the values as scalars (constant-coefficient stencils): no value loads at all
        out["uniform_values"] = s.uniform_values
        if s.uniform_values:
            _lib.call("lsk_set_option", b"sliced.uniform", 1)
            for place1 in (0, 16, 64, 256):
                out[f"uniform_place{place1}_ms"] = round(timed(ctx, lambda: be.csr_sliced_spmv(s, 0, matrix.rows, xs[0], ys[0], 0, place1), 20), 3)
            out["uniform_dot_ms"] = round(timed(ctx, lambda: be.csr_sliced_spmv_dot(s, matrix.rows, xs[0], ys[0], 0, xs[1],
                                                                                     shares, place), 20), 3)
            for place2 in (0, 64):
                out[f"uniform_pair_place{place2}_ms"] = round(timed(ctx, lambda: be.csr_sliced_spmv2(s, matrix.rows, xs, ys, 0, placement=place2), 10), 3)
            out["uniform_bytes"] = s.algorithmic_bytes(matrix.rows, esz, without_values=True)
            out["uniform_cg_ms_per_step"] = round(timed(ctx, solver.step, 30), 3)
            _lib.call("lsk_set_option", b"sliced.uniform", 0)
        planner.matvec(3, 2)   # (the steps timed above moved the vectors on: both products on what they hold now)
        a = planner.get_vector(3, 0).owned.clone()
        held, matrix.sliced_window = matrix.sliced_window, None
        out["c32_ms"] = round(timed(ctx, lambda: planner.matvec(3, 2), 20), 3)
        matrix.sliced_window = held
        b = planner.get_vector(3, 0).owned
        out["max_rel_diff_vs_c32"] = float((a - b).abs().max() / b.abs().max())
        out["cg_ms_per_step"] = round(timed(ctx, solver.step, 30), 3)
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
