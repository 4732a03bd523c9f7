"""Maximum sizes: element and row counts beyond 2^31, where 32-bit index arithmetic would wrap.
Checked against torch on the device (whole vectors) and the CPU oracle (slices around the
boundaries).  Needs ~100 GB of device memory; skipped on smaller cards."""
import ctypes

import numpy as np
import pytest
import torch

import oracle
from legionsolvers_amd import _lib

pytestmark = pytest.mark.gpu


def P(t):
    return ctypes.c_void_p(t.data_ptr())


def stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _need(gib):
    free, _ = torch.cuda.mem_get_info()
    if free < gib * (1 << 30):
        pytest.skip(f"needs {gib} GiB of free device memory")


def test_blas1_beyond_2_31_elements(gpu_ctx):
    _need(40)
    n = (1 << 31) + 1027
    print(f"case blas1 n={n} fp32", flush=True)
    g = torch.Generator(device="cuda").manual_seed(3)
    x = torch.randn(n, generator=g, device="cuda", dtype=torch.float32)
    y = torch.randn(n, generator=g, device="cuda", dtype=torch.float32)
    windows = [(0, 4096), ((1 << 31) - 2048, (1 << 31) + 1027), (n - 4096, n), ((1 << 30) + 7, (1 << 30) + 4103)]
    keep = [(x[a:b].cpu().numpy().copy(), y[a:b].cpu().numpy().copy()) for a, b in windows]
    alpha = np.float32(-0.625)
    _lib.call("lsk_axpy_f32", n, float(alpha), P(x), P(y), stream())      # y = fma(alpha, x, y)
    for (a, b), (xs, ys) in zip(windows, keep):
        want = ys.copy()
        oracle.axpy(alpha, xs, want)
        assert np.array_equal(y[a:b].cpu().numpy(), want), (a, b)
    # dot over the whole range against torch in fp64, block by block
    out = torch.zeros(1, device="cuda", dtype=torch.float32)
    ws = torch.empty(_lib.lib().lsk_dot_workspace_bytes(0), dtype=torch.uint8, device="cuda")
    _lib.call("lsk_dot_f32", n, P(x), P(y), P(out), P(ws), ws.numel(), stream())
    want, bound = 0.0, 0.0
    step = 1 << 28
    for a in range(0, n, step):
        prod = x[a:a + step].double() * y[a:a + step].double()
        want += float(prod.sum())
        bound += float(prod.abs().sum())
    assert abs(float(out) - want) <= 2e-6 * bound
    # fill and scal touch every element: the tail would stay untouched if the count wrapped
    _lib.call("lsk_fill_f32", n, 3.0, P(y), stream())
    _lib.call("lsk_scal_f32", n, 0.5, P(y), stream())
    assert float(y[-1]) == 1.5 and float(y[1 << 31]) == 1.5 and float(y[0]) == 1.5
    assert float(y.min()) == 1.5 and float(y.max()) == 1.5


def test_csr_spmv_beyond_2_31_rows(gpu_ctx):
    """A diagonal matrix with 2^31 + 517 rows (one entry per row, 256 rows per wave)."""
    _need(110)
    rows = (1 << 31) + 517
    print(f"case csr rows={rows} fp32 diagonal", flush=True)
    idx = torch.arange(rows, device="cuda", dtype=torch.int64)
    rects = torch.stack([idx, idx], dim=1).contiguous()
    val = (idx % 7 + 1).to(torch.float32)
    x = (idx % 5 + 1).to(torch.float32)
    y = torch.full((rows,), -1.0, device="cuda", dtype=torch.float32)
    _lib.call("lsk_csr_spmv_f32", rows, rows, P(rects), 0, P(idx), P(val), P(x), 0, rows - 1, P(y), 0, stream())
    assert bool((y == val * x).all())
    del rects
    # the same matrix as COO
    y.zero_()
    _lib.call("lsk_coo_spmv_f32", rows, P(idx), P(idx), P(val), P(x), 0, rows - 1, P(y), 0, rows - 1, stream())
    assert bool((y == val * x).all())


@pytest.mark.parametrize("npdt,n,tol", [(np.float64, (1 << 28) + 12345, 1e-13), (np.float32, (1 << 29) + 777, 2e-6)],
                         ids=["f64", "f32"])
def test_fused_updates_on_vectors_beyond_the_grid_caps(gpu_ctx, npdt, n, tol):
    """The fused updates on vectors of 2 GiB (2^28 + 12 345 fp64 / 2^29 + 777 fp32 elements): more 16 KiB chunks than the 65 536
    workgroups the streaming form launches at most, so every workgroup takes several; with 2048 (the least), the
    host layers' and the most partial sums on offer.  Every element against the oracle's loops bit by bit, the sums
    against fp64 sums of the oracle's results."""
    _need(30)
    be = gpu_ctx.backend
    tdt = torch.float64 if npdt == np.float64 else torch.float32
    print(f"case fused updates n={n} {np.dtype(npdt).name}", flush=True)
    g = torch.Generator(device="cuda").manual_seed(5)
    x, p, r = (torch.randn(n, generator=g, device="cuda", dtype=tdt) for _ in range(3))
    xh, ph, rh = (t.cpu().numpy() for t in (x, p, r))
    a, b = npdt(0.37), npdt(-1.3)
    ad, bd = (torch.tensor([v], dtype=tdt, device="cuda") for v in (a, b))
    # z <- fma(a, p, z) and p <- fma(b, p, r) in one pass (CG's closing updates, src/CGSolver.hpp:49-54)
    be.axpy_xpay(x, [ad], p, [bd], r)
    oracle.axpy(a, ph, xh)
    oracle.xpay(b, rh, ph)
    assert np.array_equal(x.cpu().numpy(), xh) and np.array_equal(p.cpu().numpy(), ph)
    # y <- fma(a, x, y) with the shares of y . y, and of y . w
    for slots in (2048, be.sqnorm_slots(r), 65536):
        partials = torch.empty(slots, dtype=tdt, device="cuda")
        out = torch.empty(1, dtype=tdt, device="cuda")
        count = be.axpy_sqnorm(r, [ad], p, partials)
        assert 1 <= count <= slots
        be.sum_partials(count, partials, out)
        oracle.axpy(a, ph, rh)
        assert np.array_equal(r.cpu().numpy(), rh), slots
        r64, p64 = rh.astype(np.float64), ph.astype(np.float64)
        want = float(np.dot(r64, r64))
        assert abs(float(out) - want) <= tol * want, (slots, float(out), want)
        count = be.axpy_dot(r, [bd], x, p, partials)
        be.sum_partials(count, partials, out)
        oracle.axpy(b, xh, rh)
        assert np.array_equal(r.cpu().numpy(), rh), slots
        r64 = rh.astype(np.float64)
        want, bound = float(np.dot(r64, p64)), float(np.dot(np.abs(r64), np.abs(p64)))
        assert abs(float(out) - want) <= tol * bound, (slots, float(out), want)


def test_classical_sweep_leaves_on_large_vectors(gpu_ctx):
    """lsk_multi_dot_* / lsk_multi_axpy_* (the classical Gram-Schmidt option's two passes) on fp32 vectors of
    0.6 GiB, eleven basis vectors (two groups of at most eight): the update equals the chain of axpys in ascending
    order bit by bit, with and without the shares of the new norm; the inner products equal fp64 sums to fp32
    rounding."""
    _need(20)
    be = gpu_ctx.backend
    n, k = (1 << 27) + (1 << 25) + 333, 11
    print(f"case classical sweep leaves n={n} k={k} fp32", flush=True)
    g = torch.Generator(device="cuda").manual_seed(8)
    basis = [torch.randn(n, generator=g, device="cuda", dtype=torch.float32) for _ in range(k)]
    w = torch.randn(n, generator=g, device="cuda", dtype=torch.float32)
    wh = w.cpu().numpy()
    h = torch.empty(k, dtype=torch.float32, device="cuda")
    be.multi_dot(basis, w, h)
    got_h = h.cpu().numpy()
    w64 = wh.astype(np.float64)
    for j, v in enumerate(basis):
        v64 = v.cpu().numpy().astype(np.float64)
        want, bound = float(np.dot(v64, w64)), float(np.dot(np.abs(v64), np.abs(w64)))
        assert abs(float(got_h[j]) - want) <= 2e-6 * bound, j
    for with_norm in (False, True):
        partials = torch.empty(be.sqnorm_slots(w), dtype=torch.float32, device="cuda") if with_norm else None
        count = be.multi_axpy(w, h, basis, negate=True, partials=partials)
        for j, v in enumerate(basis):
            oracle.axpy(np.float32(-got_h[j]), v.cpu().numpy(), wh)
        assert np.array_equal(w.cpu().numpy(), wh), with_norm
        if with_norm:
            out = torch.empty(1, dtype=torch.float32, device="cuda")
            be.sum_partials(count, partials, out)
            want = float(np.dot(wh.astype(np.float64), wh.astype(np.float64)))
            assert abs(float(out) - want) <= 2e-6 * want


def test_placing_the_workspace_never_costs_the_matrix_its_copy(gpu_ctx):
    """The planner holds blocks of a third of the device in front of the workspace while it allocates it
    (SquarePlanner.allocate_workspace).  With two systems on the 27-point 512^3 matrix the workspace does not fit
    behind two such blocks: the planner must then hold fewer, and no vector may be served by giving up the
    matrix's set-up-time copy."""
    import gc
    from legionsolvers_amd import Context
    from legionsolvers_amd import benchmark_stencil as bs
    gc.collect()
    torch.cuda.empty_cache()
    _need(250)
    print("case two systems 512^3: workspace placement under memory pressure", flush=True)
    ctx = Context(torch.device("cuda", 0))
    matrix, planner, solver = bs.build_problem(ctx, 4, 512, 512, 512, 1, 1, torch.float64, num_systems=2)
    assert not getattr(matrix, "setup_notes", []), matrix.setup_notes
    assert matrix.sliced, "the sliced-pattern copy was given up"
    placed = planner.workspace_placement
    assert placed is not None and 0 <= placed["blocks_in_front"] < len(placed["product_ms_by_blocks_in_front"])
    assert ctx.backend.optional_memory_stays is False
    for _ in range(3):
        solver.step()
    history = [float(h.get_value()) for h in solver.residual_norm_squared]
    # (r.r rises at the first step of this problem, on the CPU as well; two equal systems: twice one system's r.r,
    #  whose first steps on the CPU loop are 134217728, 11387733438.4979, 11234230381.4603, 11083210600.5105)
    want = [2 * v for v in (134217728.0, 11387733438.497883, 11234230381.460312, 11083210600.510529)]
    assert all(abs(h - w) <= 1e-10 * w for h, w in zip(history, want)), history
    assert matrix.sliced and not matrix.setup_notes
    del solver, planner, matrix
    gc.collect()
    torch.cuda.empty_cache()
