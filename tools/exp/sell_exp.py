#!/usr/bin/env python3
"""Experiment: the CSR product on a SELL-64 copy with 32-bit columns (tools/exp/sell_exp.hip) against the
library's product on the 32-bit structure, on the benchmark stencils.  Build the kernel library first:
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Xclang -target-feature -Xclang -packed-fp32-ops
      -shared tools/exp/sell_exp.hip -o tools/exp/libsell_exp.so"""
import ctypes, os, sys
import torch
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
os.environ["LSK_SLICED"] = "0"
from legionsolvers_amd import Context  # noqa: E402
from legionsolvers_amd import benchmark_stencil as bs  # noqa: E402


def timed(fn, reps=10):
    fn()
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(reps):
        fn()
    t1.record()
    t1.synchronize()
    return t0.elapsed_time(t1) / reps


def main():
    dim = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 512
    lib = ctypes.CDLL(os.path.join(REPO, "tools", "exp", "libsell_exp.so"))
    ctx = Context(torch.device("cuda", 0))
    shape = (n, n, n) if dim >= 3 else (n, n, 1)
    matrix, planner, solver = bs.build_problem(ctx, dim, shape[0], shape[1], shape[2], 1, 1, torch.float64, num_systems=1)
    be = ctx.backend
    rows, nnz = matrix.rows, matrix.nnz
    rowoff, col32, val = matrix.rowoff[0], matrix.col32, matrix.val
    x_lo, x_hi = matrix.col32_window
    P = lambda t: ctypes.c_void_p(t.data_ptr())  # noqa: E731
    lens = (rowoff[1:].long() & 0xffffffff) - (rowoff[:-1].long() & 0xffffffff)
    slices = (rows + 63) // 64
    padded_rows = slices * 64
    lens_p = torch.zeros(padded_rows, dtype=torch.int64, device="cuda")
    lens_p[:rows] = lens
    width4 = ((lens_p.view(slices, 64).max(dim=1).values + 3) // 4).to(torch.int32)
    incl = torch.cumsum(width4.long(), 0)
    base = (incl - width4.long()).contiguous()
    total_q = int(incl[-1])
    del lens, lens_p, incl
    padded = total_q * 256
    print(f"rows {rows} nnz {nnz} padded {padded} ({padded / nnz:.4f} x)", flush=True)
    col_out = torch.empty(padded, dtype=torch.int32, device="cuda")
    val_out = torch.empty(padded, dtype=torch.float64, device="cuda")
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    rc = lib.sell_exp_fill(ctypes.c_int64(rows), ctypes.c_int64(padded_rows), P(rowoff), P(col32), P(val), P(base),
                           P(width4), P(col_out), P(val_out), stream)
    assert rc == 0, rc
    torch.cuda.synchronize()
    g = torch.Generator(device="cuda").manual_seed(5)
    x = torch.randn(x_hi - x_lo + 1, dtype=torch.float64, device="cuda", generator=g)
    y_ref = torch.empty(rows, dtype=torch.float64, device="cuda")
    y = torch.empty(rows, dtype=torch.float64, device="cuda")
    ms = timed(lambda: be.csr_spmv(rows, nnz, rowoff, 0, col32, val, x, x_lo, x_hi, y_ref, 0, matrix.placement.get(0, 0)))
    bytes_c32 = nnz * 12 + rows * 4 + rows * 16
    print(f"c32 wave-tile kernel: {ms:.3f} ms = {bytes_c32 / ms / 1e6:.0f} GB/s of {bytes_c32 / 1e9:.2f} GB", flush=True)
    bytes_sell = padded * 12 + slices * 12 + rows * 16
    for form in range(7):
        def run():
            rc = lib.sell_exp_spmv(form, ctypes.c_int64(rows), P(base), P(width4), P(col_out), P(val_out), P(x),
                                   ctypes.c_uint32(x_hi - x_lo), P(y), 0, stream)
            assert rc == 0, rc
        y.fill_(-1.0)
        ms = timed(run)
        diff = float((y - y_ref).abs().max())
        print(f"sell form {form}: {ms:.3f} ms = {bytes_sell / ms / 1e6:.0f} GB/s of {bytes_sell / 1e9:.2f} GB   max |diff| {diff:.3e}", flush=True)


main()
