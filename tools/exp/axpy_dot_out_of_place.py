#!/usr/bin/env python3
"""Experiment (patched library tools/exp/liblsk_outofplace.so: the axpy + dot kernel stores to y + an offset set by
lsk_exp_set_out_offset): GMRES's sweep kernel y -= h x, y . w on 768^3 fp32 vectors, in place against out of place
into a vector of the same part of HBM and into one of another part.  Run with LSK_LIBRARY pointing at that library."""
import ctypes, os, sys
import torch
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
from legionsolvers_amd import Context  # noqa: E402


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(reps):
        fn()
    t1.record()
    t1.synchronize()
    return t0.elapsed_time(t1) / reps


def main():
    ctx = Context(torch.device("cuda", 0))
    be = ctx.backend
    for dtype, n, label in ((torch.float32, 768 ** 3, "768^3 fp32"), (torch.float64, 512 ** 3, "512^3 fp64")):
        a = [torch.zeros(n, dtype=dtype, device="cuda") for _ in range(4)]
        third = torch.cuda.mem_get_info()[1] // 3
        block = torch.empty(third, dtype=torch.uint8, device="cuda")
        b = [torch.zeros(n, dtype=dtype, device="cuda") for _ in range(4)]
        alpha = be.new_scalar(1e-3, dtype)
        partials = torch.zeros(1 << 16, dtype=dtype, device="cuda")
        y, x, w = b[0], b[1], b[2]

        def run(dst):
            be.lib.lsk_exp_set_out_offset(ctypes.c_longlong(dst.data_ptr() - y.data_ptr()))
            return timed(lambda: be.axpy_dot(y, [alpha], x, w, partials))
        res = {"in place": run(y), "into a fourth vector of the same part": run(b[3]), "into a vector of the other part": run(a[3])}
        be.lib.lsk_exp_set_out_offset(ctypes.c_longlong(0))
        print(label, {k: round(v, 4) for k, v in res.items()}, flush=True)
        del a, b, block, partials
        torch.cuda.empty_cache()


main()
