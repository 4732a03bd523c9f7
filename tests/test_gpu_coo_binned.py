"""GPU parity of the binned COO product (lsk_coo_binned_*) against the CPU oracle's COO loop
(oracle.coo_matvec, the restatement of src/COOMatrixTasks.cpp:58-76), through the C ABI.

The sums are formed by LDS atomics in the hardware's order, so the bar is the COO kernel's:
relative to the sum of the absolute products of the row (1e-13 fp64 / 2e-6 fp32 of it)."""
import ctypes

import numpy as np
import pytest
import torch

import oracle
from legionsolvers_amd import _lib

pytestmark = pytest.mark.gpu

DT = {"f32": (np.float32, torch.float32, 2e-6), "f64": (np.float64, torch.float64, 1e-13)}


def P(t):
    return ctypes.c_void_p(t.data_ptr())


def stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


@pytest.fixture(autouse=True)
def _init(gpu_ctx):
    yield


def compress(field, lo, hi):
    out = torch.empty(field.numel(), dtype=torch.int32, device="cuda")
    _lib.call("lsk_csr_compress_cols_u32", field.numel(), P(field), lo, hi, P(out), stream())
    return out


def make(suf, r32, c32, vd, y_first, y_last, x_extent, block_cols, tile_rows):
    handle = ctypes.c_void_p(0)
    code = getattr(_lib.lib(), f"lsk_coo_binned_create_{suf}")(
        vd.numel(), P(r32), P(c32), P(vd), y_first, y_last, x_extent, block_cols, tile_rows, stream(),
        ctypes.byref(handle))
    return code, handle


def info(handle):
    out = (ctypes.c_int64 * 8)()
    _lib.call("lsk_coo_binned_info", handle, out)
    return dict(zip(("live", "padded", "block_cols", "tile_rows", "blocks", "tiles", "units", "segments"),
                    (int(v) for v in out)))


def binned_case(suf, r, c, v, x, x_lo, x_hi, y0, y_lo, y_hi, row_origin, block_cols, tile_rows, tag):
    """One matrix, one pair of windows: beta = 1 onto y0 and beta = 0, aligned and unaligned vectors,
    then the same after the values changed in place."""
    npdt, tdt, tol = DT[suf]
    print(f"case coo_binned_{suf} {tag} nnz={v.size} x=[{x_lo},{x_hi}] y=[{y_lo},{y_hi}] origin={row_origin} "
          f"blocks of {block_cols} tiles of {tile_rows}", flush=True)
    xs = np.ascontiguousarray(x[x_lo:x_hi + 1])
    rows = y_hi - y_lo + 1
    rd, cd, vd = dev(r), dev(c), dev(v)
    r32 = compress(rd, row_origin, row_origin + 2 ** 20)
    c32 = compress(cd, x_lo, x_hi)
    code, handle = make(suf, r32, c32, vd, y_lo - row_origin, y_hi - row_origin, x_hi - x_lo + 1, block_cols,
                        tile_rows)
    assert code == _lib.LSK_OK, (code, _lib.lib().lsk_last_error())
    try:
        inside = (r >= y_lo) & (r <= y_hi) & (c >= x_lo) & (c <= x_hi)
        got = info(handle)
        assert got["live"] == int(inside.sum()), got
        assert got["padded"] % 8 == 0 and got["padded"] >= got["live"]
        for values in (v, (v * npdt(-0.5) + npdt(0.25)).astype(npdt)):
            if values is not v:
                vd.copy_(dev(values))  # in place: the caller's array, new values
                _lib.call(f"lsk_coo_binned_update_values_{suf}", handle, values.size, P(vd), stream())
            bound = np.zeros(rows, dtype=np.float64)
            oracle.coo_matvec(r, c, np.abs(values).astype(np.float64), np.abs(xs).astype(np.float64), x_lo, x_hi,
                              bound, y_lo, y_hi)
            for beta in (1, 0):
                start = y0[y_lo:y_hi + 1].copy() if beta else np.zeros(rows, dtype=npdt)
                want = start.copy()
                oracle.coo_matvec(r, c, values, xs, x_lo, x_hi, want, y_lo, y_hi)
                lim = tol * (bound + np.abs(start) + np.abs(want) + 1e-300)
                for shift in (0, 1):  # 16-byte aligned vectors, and vectors one element off
                    xbuf = torch.zeros(xs.size + 4, dtype=tdt, device="cuda")
                    ybuf = torch.full((rows + 4,), 7.0, dtype=tdt, device="cuda")
                    xv, yv = xbuf[shift:shift + xs.size], ybuf[shift:shift + rows]
                    xv.copy_(dev(xs))
                    yv.copy_(dev(y0[y_lo:y_hi + 1]))
                    _lib.call(f"lsk_coo_binned_spmv_{suf}", handle, P(xv), P(yv), beta, stream())
                    err = np.abs(yv.cpu().numpy().astype(np.float64) - want)
                    assert np.all(err <= lim), (tag, beta, shift, float(err.max()), int(np.argmax(err - lim)))
                    # nothing outside the rows was written
                    outside = torch.cat((ybuf[:shift], ybuf[shift + rows:]))
                    assert bool((outside == 7.0).all()), (tag, beta, shift)
    finally:
        _lib.call("lsk_coo_binned_destroy", handle)


@pytest.mark.parametrize("suf", ["f32", "f64"])
def test_coo_binned_random_matrices(suf):
    npdt, tdt, tol = DT[suf]
    rng = np.random.default_rng(61)
    n_rows, n_cols, nnz = 3000, 2500, 60000
    r = rng.integers(0, n_rows, nnz).astype(np.int64)
    c = rng.integers(0, n_cols, nnz).astype(np.int64)
    v = rng.standard_normal(nnz).astype(npdt)
    x = rng.standard_normal(n_cols).astype(npdt)
    y0 = rng.standard_normal(n_rows).astype(npdt)
    # many blocks and tiles; ragged last block and tile; one block and one tile (the defaults)
    for block_cols, tile_rows in ((256, 512), (1000, 104), (8, 8), (0, 0)):
        if block_cols == 8 and suf == "f32":
            continue  # 375 x 313 bins of a dozen entries: once is enough
        binned_case(suf, r, c, v, x, 0, n_cols - 1, y0, 0, n_rows - 1, 0, block_cols, tile_rows, "whole")
    # windows inside the matrix: entries outside either are left out; rows counted from an origin
    binned_case(suf, r, c, v, x, 300, 2100, y0, 500, 2499, 0, 256, 512, "windows")
    binned_case(suf, r, c, v, x, 17, 2400, y0, 1001, 1900, 400, 512, 256, "windows+origin")
    binned_case(suf, r, c, v, x, 0, n_cols - 1, y0, 7, 7, 0, 64, 64, "one-row")
    # row-sorted entries (what a COO piece usually is) and duplicates of one position
    order = np.argsort(r, kind="stable")
    binned_case(suf, r[order], c[order], v[order], x, 0, n_cols - 1, y0, 0, n_rows - 1, 0, 256, 512, "row-sorted")
    rd = np.concatenate((r, np.full(5000, 3, dtype=np.int64)))
    cd = np.concatenate((c, rng.integers(0, 10, 5000).astype(np.int64)))
    vd = np.concatenate((v, rng.standard_normal(5000).astype(npdt)))
    binned_case(suf, rd, cd, vd, x, 0, n_cols - 1, y0, 0, n_rows - 1, 0, 256, 512, "one-long-row")


@pytest.mark.parametrize("suf", ["f32", "f64"])
def test_coo_binned_units_longer_than_one_workgroup_run(suf):
    """A block with more entries than one workgroup of the first pass streams (2^17), so that it is
    cut into several, and segments of thousands of entries for the second pass."""
    npdt, tdt, tol = DT[suf]
    rng = np.random.default_rng(62)
    n_rows, n_cols, nnz = 40000, 20000, 700000
    r = rng.integers(0, n_rows, nnz).astype(np.int64)
    c = rng.integers(0, n_cols, nnz).astype(np.int64)
    v = rng.standard_normal(nnz).astype(npdt)
    x = rng.standard_normal(n_cols).astype(npdt)
    y0 = rng.standard_normal(n_rows).astype(npdt)
    binned_case(suf, r, c, v, x, 0, n_cols - 1, y0, 0, n_rows - 1, 0, 0, 0, "defaults")
    binned_case(suf, r, c, v, x, 0, n_cols - 1, y0, 0, n_rows - 1, 0, 8192, 4096, "8192x4096")


def test_coo_binned_stencil_and_nonfinite_x():
    """A stencil in COO form (nearly all bins empty), and an infinite x entry: it reaches exactly
    the rows that have an entry in its column, as in the CPU loop -- the pads never multiply it."""
    offs, cfs = oracle.benchmark_stencil_offsets(4)
    _, row, col, val = oracle.linearized_stencil((12, 11, 10), offs, cfs, dtype=np.float64, coo=True)
    n = 12 * 11 * 10
    rng = np.random.default_rng(63)
    x = rng.standard_normal(n)
    y0 = rng.standard_normal(n)
    binned_case("f64", row, col, val, x, 0, n - 1, y0, 0, n - 1, 0, 128, 128, "stencil")
    x[333] = np.inf
    want = np.zeros(n)
    with np.errstate(invalid="ignore"):
        oracle.coo_matvec(row, col, val, x, 0, n - 1, want, 0, n - 1)
    rd, cd, vd = dev(row), dev(col), dev(val)
    code, handle = make("f64", compress(rd, 0, n - 1), compress(cd, 0, n - 1), vd, 0, n - 1, n, 128, 128)
    assert code == _lib.LSK_OK
    try:
        yd = torch.zeros(n, dtype=torch.float64, device="cuda")
        _lib.call("lsk_coo_binned_spmv_f64", handle, P(dev(x)), P(yd), 0, stream())
        got = yd.cpu().numpy()
        assert np.array_equal(np.isfinite(got), np.isfinite(want))
        assert np.array_equal(got[~np.isfinite(want)], want[~np.isfinite(want)], equal_nan=True)
    finally:
        _lib.call("lsk_coo_binned_destroy", handle)


def test_coo_binned_declines_and_argument_errors():
    lib = _lib.lib()
    r32 = torch.tensor([0, 1, 2], dtype=torch.int32, device="cuda")
    c32 = torch.tensor([5, 6, 7], dtype=torch.int32, device="cuda")
    vd = torch.ones(3, dtype=torch.float64, device="cuda")
    # every entry outside the x window: nothing to multiply, the caller keeps its other form
    code, handle = make("f64", r32, c32, vd, 0, 2, 4, 0, 0)
    assert code == _lib.LSK_ERR_UNSUPPORTED and not handle.value
    # bad block sizes
    for bc, tr in ((12, 8), (8, 20480), (-8, 8)):
        code, handle = make("f64", r32, c32, vd, 0, 2, 8, bc, tr)
        assert code == _lib.LSK_ERR_ARG and not handle.value, (bc, tr)
    code, handle = make("f64", r32, c32, vd, 0, 2, 8, 8, 8)
    assert code == _lib.LSK_OK
    try:
        x = torch.arange(8, dtype=torch.float64, device="cuda")
        y = torch.zeros(3, dtype=torch.float64, device="cuda")
        assert lib.lsk_coo_binned_spmv_f64(handle, P(x), P(y), 2, stream()) == _lib.LSK_ERR_ARG
        assert lib.lsk_coo_binned_spmv_f32(handle, P(x), P(y), 0, stream()) == _lib.LSK_ERR_ARG
        assert lib.lsk_coo_binned_update_values_f64(handle, 2, P(vd), stream()) == _lib.LSK_ERR_ARG
        _lib.call("lsk_coo_binned_spmv_f64", handle, P(x), P(y), 0, stream())
        assert y.tolist() == [5.0, 6.0, 7.0]
    finally:
        _lib.call("lsk_coo_binned_destroy", handle)


def test_coo_binned_two_live_copies_of_different_sizes():
    """The kernels' limit on dynamic LDS belongs to the kernel, not to a copy: a copy with the largest
    blocks and tiles (160 KiB of LDS in both passes), then a second live copy with small ones, then a
    product on the large one again -- and a block count beyond what one capped set-up grid covers."""
    rng = np.random.default_rng(64)
    n_rows, n_cols, nnz = 50000, 45000, 400000
    r = rng.integers(0, n_rows, nnz).astype(np.int64)
    c = rng.integers(0, n_cols, nnz).astype(np.int64)
    v = rng.standard_normal(nnz)
    x = rng.standard_normal(n_cols)
    want = np.zeros(n_rows)
    oracle.coo_matvec(r, c, v, x, 0, n_cols - 1, want, 0, n_rows - 1)
    bound = np.zeros(n_rows)
    oracle.coo_matvec(r, c, np.abs(v), np.abs(x), 0, n_cols - 1, bound, 0, n_rows - 1)
    rd, cd, vd, xd = dev(r), dev(c), dev(v), dev(x)
    r32, c32 = compress(rd, 0, n_rows - 1), compress(cd, 0, n_cols - 1)
    print("case coo_binned two live copies: 20472 x 20472, then 64 x 64", flush=True)
    code, large = make("f64", r32, c32, vd, 0, n_rows - 1, n_cols, 20472, 20472)
    assert code == _lib.LSK_OK, _lib.lib().lsk_last_error()
    code, small = make("f64", r32, c32, vd, 0, n_rows - 1, n_cols, 64, 64)
    assert code == _lib.LSK_OK, _lib.lib().lsk_last_error()
    try:
        for handle in (large, small, large):
            yd = torch.full((n_rows,), 3.0, dtype=torch.float64, device="cuda")
            _lib.call("lsk_coo_binned_spmv_f64", handle, P(xd), P(yd), 0, stream())
            err = np.abs(yd.cpu().numpy() - want)
            assert np.all(err <= 1e-13 * (bound + 1e-300)), float(err.max())
    finally:
        _lib.call("lsk_coo_binned_destroy", small)
        _lib.call("lsk_coo_binned_destroy", large)
    # blocks of 8 columns over a wide window with one tile: 2^23 + 1 blocks, more than the 2^23 threads of
    # the capped set-up grid
    n_cols2 = 8 * (1 << 23) + 8
    nnz2 = 20000
    r2 = rng.integers(0, 100, nnz2).astype(np.int64)
    c2 = np.sort(rng.integers(0, n_cols2, nnz2)).astype(np.int64)
    c2[-3:] = n_cols2 - 1 - np.arange(3)             # entries in the very last block
    v2 = rng.standard_normal(nnz2)
    print(f"case coo_binned {n_cols2 // 8} blocks of 8 columns", flush=True)
    x2 = torch.randn(n_cols2, dtype=torch.float64, device="cuda")
    code, wide = make("f64", compress(dev(r2), 0, 99), compress(dev(c2), 0, n_cols2 - 1), dev(v2), 0, 99, n_cols2, 8, 104)
    assert code == _lib.LSK_OK, _lib.lib().lsk_last_error()
    try:
        assert info(wide)["blocks"] == n_cols2 // 8 and info(wide)["live"] == nnz2
        yd = torch.zeros(100, dtype=torch.float64, device="cuda")
        _lib.call("lsk_coo_binned_spmv_f64", wide, P(x2), P(yd), 0, stream())
        xh = x2.cpu().numpy()
        want2 = np.zeros(100)
        oracle.coo_matvec(r2, c2, v2, xh, 0, n_cols2 - 1, want2, 0, 99)
        bound2 = np.zeros(100)
        oracle.coo_matvec(r2, c2, np.abs(v2), np.abs(xh), 0, n_cols2 - 1, bound2, 0, 99)
        assert np.all(np.abs(yd.cpu().numpy() - want2) <= 1e-13 * (bound2 + 1e-300))
    finally:
        _lib.call("lsk_coo_binned_destroy", wide)


@pytest.mark.parametrize("suf", ["f32", "f64"])
def test_coo_binned_product_in_two_phases(suf):
    """lsk_coo_binned_spmv_phase_*: phase 0 (the blocks of columns wholly inside an interval, reading x only
    there) then phase 1 (the rest, and the second pass) give the whole product -- whatever
    the interval, with x outside it still missing (NaN) during phase 0."""
    npdt, tdt, tol = DT[suf]
    rng = np.random.default_rng(65)
    n_rows, n_cols, nnz = 5000, 4100, 90000
    r = rng.integers(0, n_rows, nnz).astype(np.int64)
    c = rng.integers(0, n_cols, nnz).astype(np.int64)
    v = rng.standard_normal(nnz).astype(npdt)
    x = rng.standard_normal(n_cols).astype(npdt)
    rd, cd, vd, xd = dev(r), dev(c), dev(v), dev(x)
    code, handle = make(suf, compress(rd, 0, n_rows - 1), compress(cd, 0, n_cols - 1), vd, 0, n_rows - 1, n_cols, 256, 512)
    assert code == _lib.LSK_OK
    try:
        whole = torch.full((n_rows,), 3.0, dtype=tdt, device="cuda")
        _lib.call(f"lsk_coo_binned_spmv_{suf}", handle, P(xd), P(whole), 0, stream())
        # intervals: a rank's share in the middle, from the start, to the end (the ragged last block), inside one
        # block (nothing to do early), empty, everything, and reaching outside the window
        for lo, hi in ((1024, 2999), (0, 1500), (2000, n_cols - 1), (300, 400), (5, 4), (0, n_cols - 1), (-50, n_cols + 50)):
            for beta in (0, 1):
                print(f"case coo_binned phases {suf} columns [{lo},{hi}] beta={beta}", flush=True)
                y = torch.full((n_rows,), 3.0, dtype=tdt, device="cuda")
                early = torch.full_like(xd, float("nan"))
                a, b = max(lo, 0), min(hi, n_cols - 1)
                if b >= a:
                    early[a:b + 1] = xd[a:b + 1]
                _lib.call(f"lsk_coo_binned_spmv_phase_{suf}", handle, P(early), P(y), beta, 0, lo, hi, stream())
                early.copy_(xd)                                  # the rest of x "arrives"
                _lib.call(f"lsk_coo_binned_spmv_phase_{suf}", handle, P(early), P(y), beta, 1, lo, hi, stream())
                # (the first pass writes the same products either way; the second adds them with LDS atomics in
                # the hardware's order, so two runs of the same product agree to rounding, like the plain entry point)
                want = whole + 3.0 if beta else whole
                err = (y.double() - want.double()).abs().max().item()
                assert err <= 100 * tol * float(whole.abs().max()), (lo, hi, err)
                assert not bool(torch.isnan(y).any())
        lib = _lib.lib()
        y = torch.zeros(n_rows, dtype=tdt, device="cuda")
        assert getattr(lib, f"lsk_coo_binned_spmv_phase_{suf}")(handle, P(xd), P(y), 0, 2, 0, 10, stream()) == _lib.LSK_ERR_ARG
        assert getattr(lib, f"lsk_coo_binned_spmv_phase_{suf}")(handle, P(xd), P(y), 0, -1, 0, 10, stream()) == _lib.LSK_ERR_ARG
    finally:
        _lib.call("lsk_coo_binned_destroy", handle)
