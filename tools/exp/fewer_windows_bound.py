#!/usr/bin/env python3
"""Experiment (timing only, results of the patched kernels are wrong): the 27-point 512^3 fp64 sliced product, single
and two-vector, with the last 3 or 5 of each wave's 9 windows of x not staged (tools/exp/liblsk_skip3.so, _skip5.so:
sliced_spmv.hip built with -DLSK_EXP_SKIP_RUNS=3|5 in stage()), against the library's kernel IN THE SAME PROCESS on
the same copy and vectors: what fewer window bytes per row could gain at most."""
import ctypes, os, sys
import torch
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
from legionsolvers_amd import Context  # noqa: E402
from legionsolvers_amd import benchmark_stencil as bs  # noqa: E402


def timed(fn, reps=10):
    fn(); fn()
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(reps):
        fn()
    t1.record()
    t1.synchronize()
    return t0.elapsed_time(t1) / reps


def main():
    ctx = Context(torch.device("cuda", 0))
    matrix, planner, solver = bs.build_problem(ctx, 4, 512, 512, 512, 1, 1, torch.float64, num_systems=2)
    rows = matrix.rows
    handle = matrix.sliced[0]
    P = lambda t: ctypes.c_void_p(t.data_ptr())  # noqa: E731
    xa, xb = planner.get_vector(2, 0)._data, planner.get_vector(2, 1)._data
    ya, yb = planner.get_vector(3, 0)._data, planner.get_vector(3, 1)._data
    assert xa.numel() == rows, (xa.numel(), rows)
    print("workspace placement:", planner.workspace_placement, flush=True)
    libs = {"library": ctx.backend.lib}
    for name in ("skip3", "skip5"):
        lib = ctypes.CDLL(os.path.join(REPO, "tools", "exp", f"liblsk_{name}.so"))
        assert lib.lsk_init(0) == 0
        libs[name] = lib
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    for rnd in range(2):
        for name, lib in libs.items():
            out = {}
            for place in (2, 64):
                assert lib.lsk_set_option(b"xcd_remap", place) == 0
                single = lambda: lib.lsk_csr_sliced_spmv_f64(handle.handle, ctypes.c_int64(0), ctypes.c_int64(rows), P(xa), P(ya), 0, stream)  # noqa: E731
                pair = lambda: lib.lsk_csr_sliced_spmv2_f64(handle.handle, ctypes.c_int64(0), ctypes.c_int64(rows), P(xa), P(xb), P(ya), P(yb), 0, stream)  # noqa: E731
                assert single() == 0 and pair() == 0
                out[place] = (round(timed(single), 3), round(timed(pair), 3))
            print(f"round {rnd} {name}: (single, pair) ms by runs of blocks per XCD: {out}", flush=True)
        ctx.backend.lib.lsk_set_option(b"xcd_remap", 0)


main()
