"""GPU parity tests of the CSR / COO SpMV kernels and the device-side set-up helpers, called
through the C ABI and compared with the CPU oracle.

Index work (generators, rect rebasing, image bounds) is bit-exact.  SpMV results are sums
whose order differs from the CPU's ascending-k loop: tolerance 1e-13 (fp64) / 1e-5 (fp32)
relative to sum_k |val_k x_col_k| of the row.
"""
import ctypes

import numpy as np
import pytest
import torch

import oracle
from legionsolvers_amd import _lib

pytestmark = pytest.mark.gpu

DT = {"f32": (np.float32, torch.float32, 1e-5), "f64": (np.float64, torch.float64, 1e-13)}
BLOCK_ROWS = {"f32": 1024, "f64": 512}  # rows per block of the sliced-pattern structure


def P(t):
    return ctypes.c_void_p(t.data_ptr() if t is not None else 0)


def stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


@pytest.fixture(scope="module", autouse=True)
def _init(gpu_ctx):
    return gpu_ctx


@pytest.fixture(params=[1, 0], ids=["wave-tiles", "workgroup-tiles"])
def csr_variant(request):
    """All forms of the CSR kernel (option csr.variant); the default is restored afterwards."""
    default = ctypes.c_int(0)
    _lib.call("lsk_get_option", b"csr.variant", ctypes.byref(default))
    _lib.call("lsk_set_option", b"csr.variant", request.param)
    yield request.param
    _lib.call("lsk_set_option", b"csr.variant", default.value)


def csr_case(suf, rects, k_base, col, val, x, x_lo, x_hi, beta=0, y0=None, tag=""):
    """Run lsk_csr_spmv on the given piece and compare with the oracle."""
    npdt, tdt, tol = DT[suf]
    rows = rects.shape[0]
    print(f"case csr_spmv_{suf} {tag} rows={rows} nnz={val.size} k_base={k_base} x=[{x_lo},{x_hi}] beta={beta}",
          flush=True)
    y0 = np.zeros(rows, dtype=npdt) if y0 is None else y0.astype(npdt)
    want = y0.copy()
    oracle.csr_matvec(rects, k_base, col, val, x, x_lo, x_hi, want, beta=beta)
    bound = np.zeros(rows, dtype=npdt)
    oracle.csr_matvec(rects, k_base, col, np.abs(val), np.abs(x), x_lo, x_hi, bound, beta=0)
    rd, cd, vd, xd, yd = dev(rects), dev(col), dev(val), dev(x), dev(y0)
    _lib.call(f"lsk_csr_spmv_{suf}", rows, val.size, P(rd), k_base, P(cd), P(vd), P(xd), x_lo, x_hi,
              P(yd), beta, stream())
    got = yd.cpu().numpy()
    err = np.abs(got.astype(np.float64) - want.astype(np.float64))
    lim = tol * (bound.astype(np.float64) + np.abs(y0) + 1e-300)
    assert np.all(err <= lim), f"{tag}: max err {err.max()} at row {int(np.argmax(err - lim))}"
    # the same piece with compressed (32-bit, window-relative) columns: the compression is bit-exact,
    # and the product is bit-identical to the 64-bit wave-tile form
    c32 = torch.empty(val.size, dtype=torch.int32, device="cuda")
    _lib.call("lsk_csr_compress_cols_u32", val.size, P(cd), x_lo, x_hi, P(c32), stream())
    inside = (col >= x_lo) & (col <= x_hi)
    assert np.array_equal(c32.cpu().numpy().view(np.uint32),
                          np.where(inside, col - x_lo, 0xFFFFFFFF).astype(np.uint32)), f"{tag}: compression"
    # ... and the rects as rows + 1 offsets, when they are back to back inside the piece
    back_to_back = bool(np.all(rects[1:, 0] == rects[:-1, 1] + 1) and np.all(rects[:, 1] + 1 >= rects[:, 0])
                        and rects[0, 0] >= k_base and rects[-1, 1] + 1 - k_base <= val.size) if rows else True
    off32 = torch.full((rows + 1,), -7, dtype=torch.int32, device="cuda")
    code = _lib.lib().lsk_csr_compress_rows_u32(rows, val.size, P(rd), k_base, P(off32), stream())
    assert code == (_lib.LSK_OK if back_to_back else _lib.LSK_ERR_UNSUPPORTED), f"{tag}: row compression said {code}"
    if not back_to_back:
        return got
    want_off = np.concatenate([rects[:, 0] - k_base, rects[-1:, 1] + 1 - k_base]) if rows else np.zeros(1)
    assert np.array_equal(off32.cpu().numpy().view(np.uint32), want_off.astype(np.uint32)), f"{tag}: row offsets"
    yc = dev(y0)
    _lib.call(f"lsk_csr_spmv_c32_{suf}", rows, val.size, P(off32), P(c32), P(vd), P(xd), x_hi - x_lo + 1,
              P(yc), beta, stream())
    got32 = yc.cpu().numpy()
    variant = ctypes.c_int(0)
    _lib.call("lsk_get_option", b"csr.variant", ctypes.byref(variant))
    if variant.value == 1:
        assert np.array_equal(got32, got), f"{tag}: the compressed structure changes the result"
    else:
        err = np.abs(got32.astype(np.float64) - want.astype(np.float64))
        assert np.all(err <= lim), f"{tag} (compressed structure): max err {err.max()}"
    return got


@pytest.mark.parametrize("suf", ["f32", "f64"])
@pytest.mark.parametrize("flag,extent", [(1, (1000,)), (2, (37, 41)), (3, (9, 11, 13)), (4, (12, 10, 14)),
                                         (4, (3, 3, 3)), (1, (1,)), (1, (2,))])
def test_csr_spmv_stencils(suf, flag, extent, csr_variant):
    npdt = DT[suf][0]
    offs, cfs = oracle.benchmark_stencil_offsets(flag)
    rects, _, col, val = oracle.linearized_stencil(extent, offs, cfs, dtype=npdt)
    n = rects.shape[0]
    x = np.random.default_rng(21).standard_normal(n).astype(npdt)
    csr_case(suf, rects, 0, col, val, x, 0, n - 1, tag=f"stencil{flag}{extent}")
    y0 = np.random.default_rng(22).standard_normal(n).astype(npdt)
    csr_case(suf, rects, 0, col, val, x, 0, n - 1, beta=1, y0=y0, tag="beta1")


@pytest.mark.parametrize("suf", ["f32", "f64"])
def test_csr_spmv_pieces_and_ghost_windows(suf, csr_variant):
    """Row-block pieces of a larger matrix: k_base != 0, odd first entry (so the 16-byte
    loads start off-phase), x handed over as the ghost window only."""
    npdt = DT[suf][0]
    offs, cfs = oracle.benchmark_stencil_offsets(4)
    rects, _, col, val = oracle.linearized_stencil((7, 9, 11), offs, cfs, dtype=npdt)
    n = rects.shape[0]
    x = np.random.default_rng(23).standard_normal(n).astype(npdt)
    for r0, r1 in [(0, 99), (100, 356), (357, 357), (358, n - 1), (5, 5 + 255), (17, 17 + 300)]:
        piece = np.ascontiguousarray(rects[r0:r1 + 1])
        k_lo, k_hi = int(piece[0, 0]), int(piece[-1, 1])
        c, v = np.ascontiguousarray(col[k_lo:k_hi + 1]), np.ascontiguousarray(val[k_lo:k_hi + 1])
        g_lo, g_hi = int(c.min()), int(c.max())
        csr_case(suf, piece, k_lo, c, v, np.ascontiguousarray(x[g_lo:g_hi + 1]), g_lo, g_hi,
                 tag=f"piece[{r0},{r1}]")
    # a window narrower than the image: out-of-window entries are skipped, not read
    piece = np.ascontiguousarray(rects[200:400])
    k_lo, k_hi = int(piece[0, 0]), int(piece[-1, 1])
    csr_case(suf, piece, k_lo, np.ascontiguousarray(col[k_lo:k_hi + 1]), np.ascontiguousarray(val[k_lo:k_hi + 1]),
             np.ascontiguousarray(x[250:350]), 250, 349, tag="narrow-window")


@pytest.mark.parametrize("suf", ["f32", "f64"])
def test_csr_spmv_general_rects(suf, csr_variant):
    """Rects that do not tile the piece front to back (gaps, permuted rows, empty rows, rows
    longer than the LDS buffer): the per-wave path, same results."""
    npdt = DT[suf][0]
    rng = np.random.default_rng(24)
    # random matrix, variable row lengths incl. empty rows and one very long row
    lens = rng.integers(0, 40, size=700)
    lens[5] = 0
    lens[300] = 20000
    lens[699] = 0
    n_cols = 5000
    starts = np.concatenate([[0], np.cumsum(lens)])
    nnz = int(starts[-1])
    col = rng.integers(0, n_cols, size=nnz).astype(np.int64)
    val = rng.standard_normal(nnz).astype(npdt)
    x = rng.standard_normal(n_cols).astype(npdt)
    rects = np.stack([starts[:-1], starts[1:] - 1], axis=1).astype(np.int64)
    csr_case(suf, rects, 0, col, val, x, 0, n_cols - 1, tag="variable-rows")
    # permuted rows: each row still owns its own run, but runs are visited out of order
    perm = rng.permutation(700)
    csr_case(suf, np.ascontiguousarray(rects[perm]), 0, col, val, x, 0, n_cols - 1, tag="permuted-rows")
    # gaps: every row ignores its last entry
    gap = rects.copy()
    gap[lens > 1, 1] -= 1
    csr_case(suf, gap, 0, col, val, x, 0, n_cols - 1, tag="gapped-rows")
    # all rows empty
    empty = np.stack([np.arange(50), np.arange(50) - 1], axis=1).astype(np.int64)
    got = csr_case(suf, empty, 0, col[:60], val[:60], x, 0, n_cols - 1, tag="all-empty")
    assert np.all(got == 0)
    # short rows (several rects per lane) with a few long ones in between: tiles that are taken
    # in batches, and rows that alone exceed the on-chip buffer
    for seed, (short_hi, long_len, every) in enumerate([(6, 3000, 97), (3, 700, 41), (8, 1200, 300)]):
        rng2 = np.random.default_rng(100 + seed)
        lens2 = rng2.integers(0, short_hi, size=3000)
        lens2[::every] = long_len
        starts2 = np.concatenate([[0], np.cumsum(lens2)])
        col2 = rng2.integers(0, n_cols, size=int(starts2[-1])).astype(np.int64)
        val2 = rng2.standard_normal(int(starts2[-1])).astype(npdt)
        rects2 = np.stack([starts2[:-1], starts2[1:] - 1], axis=1).astype(np.int64)
        csr_case(suf, rects2, 0, col2, val2, x, 0, n_cols - 1, tag=f"short-and-long-rows-{seed}")
    # clusters of long rows among short ones: the mean picks large tiles, the clusters overflow them
    lens3 = np.full(4000, 2)
    for c0 in range(100, 4000, 500):
        lens3[c0:c0 + 40] = 200
    starts3 = np.concatenate([[0], np.cumsum(lens3)])
    col3 = rng.integers(0, n_cols, size=int(starts3[-1])).astype(np.int64)
    val3 = rng.standard_normal(int(starts3[-1])).astype(npdt)
    rects3 = np.stack([starts3[:-1], starts3[1:] - 1], axis=1).astype(np.int64)
    csr_case(suf, rects3, 0, col3, val3, x, 0, n_cols - 1, tag="clustered-long-rows")
    csr_case(suf, rects3, 0, col3, val3, x, 0, n_cols - 1, beta=1,
             y0=rng.standard_normal(4000).astype(npdt), tag="clustered-long-rows-beta1")
    # dense-ish rows (~512 per row): few rows per workgroup
    lens = np.full(64, 512)
    starts = np.concatenate([[0], np.cumsum(lens)])
    col = rng.integers(0, n_cols, size=int(starts[-1])).astype(np.int64)
    val = rng.standard_normal(int(starts[-1])).astype(npdt)
    rects = np.stack([starts[:-1], starts[1:] - 1], axis=1).astype(np.int64)
    csr_case(suf, rects, 0, col, val, x, 0, n_cols - 1, tag="rows-of-512")


@pytest.mark.parametrize("suf", ["f32", "f64"])
@pytest.mark.parametrize("flag,extent", [(4, (64, 64, 72)), (3, (100, 100, 101)), (2, (1500, 700))])
def test_csr_spmv_many_tiles(suf, flag, extent, csr_variant):
    """Large pieces (tens of thousands of tiles), some tiles irregular in the middle of the sweep
    (rows with a gap, which take the per-row path), and a piece that starts at an odd entry."""
    npdt = DT[suf][0]
    offs, cfs = oracle.benchmark_stencil_offsets(flag)
    rects, _, col, val = oracle.linearized_stencil(extent, offs, cfs, dtype=npdt)
    n = rects.shape[0]
    x = np.random.default_rng(28).standard_normal(n).astype(npdt)
    csr_case(suf, rects, 0, col, val, x, 0, n - 1, tag=f"many-tiles{flag}{extent}")
    gap = rects.copy()
    for r in range(1000, n, 50021):  # every so often a row drops its last entry
        gap[r, 1] -= 1
    csr_case(suf, gap, 0, col, val, x, 0, n - 1, tag="many-tiles-gaps")
    r0 = 12345
    piece = np.ascontiguousarray(rects[r0:])
    k_lo = int(piece[0, 0])
    c, v = np.ascontiguousarray(col[k_lo:]), np.ascontiguousarray(val[k_lo:])
    g_lo, g_hi = int(c.min()), int(c.max())
    csr_case(suf, piece, k_lo, c, v, np.ascontiguousarray(x[g_lo:g_hi + 1]), g_lo, g_hi, beta=1,
             y0=np.random.default_rng(29).standard_normal(n - r0).astype(npdt), tag=f"many-tiles-piece k_lo={k_lo}")


@pytest.mark.parametrize("suf", ["f32", "f64"])
def test_csr_spmv_misaligned_arrays(suf, csr_variant):
    """col / val handed over at addresses that are not 16-byte aligned."""
    npdt, tdt, tol = DT[suf]
    offs, cfs = oracle.benchmark_stencil_offsets(3)
    rects, _, col, val = oracle.linearized_stencil((6, 7, 8), offs, cfs, dtype=npdt)
    n = rects.shape[0]
    x = np.random.default_rng(25).standard_normal(n).astype(npdt)
    want = np.zeros(n, dtype=npdt)
    oracle.csr_matvec(rects, 0, col, val, x, 0, n - 1, want)
    for coff in (0, 1):
        for voff in range(0, 4 if suf == "f32" else 2):
            print(f"case csr misaligned {suf} coff={coff} voff={voff}", flush=True)
            cd = dev(np.concatenate([np.zeros(coff, np.int64), col]))
            vd = dev(np.concatenate([np.zeros(voff, npdt), val]))
            xd, rd = dev(x), dev(rects)
            yd = torch.zeros(n, dtype=tdt, device="cuda")
            _lib.call(f"lsk_csr_spmv_{suf}", n, val.size, P(rd), 0, P(cd[coff:]), P(vd[voff:]), P(xd),
                      0, n - 1, P(yd), 0, stream())
            np.testing.assert_allclose(yd.cpu().numpy(), want, rtol=0, atol=tol * 20)


@pytest.mark.parametrize("suf", ["f32", "f64"])
def test_csr_spmv_c32_phases(suf):
    """Compressed columns with col32 / val handed over at every combination of 16-byte phases,
    pieces starting at every entry offset mod 4, fused dot and tuner on the compressed form."""
    npdt, tdt, tol = DT[suf]
    offs, cfs = oracle.benchmark_stencil_offsets(4)
    rects, _, col, val = oracle.linearized_stencil((9, 10, 11), offs, cfs, dtype=npdt)
    n = rects.shape[0]
    x = np.random.default_rng(61).standard_normal(n).astype(npdt)
    xd, rd = dev(x), dev(rects)
    ref = torch.zeros(n, dtype=tdt, device="cuda")
    col64, val0 = dev(col), dev(val)
    _lib.call(f"lsk_csr_spmv_{suf}", n, val.size, P(rd), 0, P(col64), P(val0), P(xd), 0, n - 1, P(ref), 0,
              stream())
    torch.cuda.synchronize()
    c32 = (col.astype(np.uint32)).view(np.int32)
    off = dev(np.concatenate([rects[:, 0], rects[-1:, 1] + 1]).astype(np.uint32).view(np.int32))
    for coff in range(4):
        for voff in range(0, 4 if suf == "f32" else 2):
            print(f"case csr c32 phases {suf} coff={coff} voff={voff}", flush=True)
            cd = dev(np.concatenate([np.zeros(coff, np.int32), c32]))
            vd = dev(np.concatenate([np.zeros(voff, npdt), val]))
            yd = torch.zeros(n, dtype=tdt, device="cuda")
            _lib.call(f"lsk_csr_spmv_c32_{suf}", n, val.size, P(off), P(cd[coff:]), P(vd[voff:]), P(xd), n,
                      P(yd), 0, stream())
            assert torch.equal(yd, ref)
    # runs of rows that start at rows 1..5 (first entry at any offset): offsets handed over from that row on
    cd, vd = dev(c32), dev(val)
    for r0 in range(1, 6):
        k_lo = int(rects[r0, 0])
        rows = n - r0 - 3
        nnz = int(rects[r0 + rows - 1, 1]) - k_lo + 1
        print(f"case csr c32 piece {suf} r0={r0} k_lo={k_lo}", flush=True)
        yd = torch.zeros(rows, dtype=tdt, device="cuda")
        _lib.call(f"lsk_csr_spmv_c32_{suf}", rows, val.size, P(off[r0:]), P(cd), P(vd), P(xd), n,
                  P(yd), 0, stream())
        assert torch.equal(yd, ref[r0:r0 + rows])
    # fused product + dot on the compressed form: same y, same partial sums as the 64-bit form
    w = dev(np.random.default_rng(62).standard_normal(n).astype(npdt))
    slots = int(_lib.lib().lsk_csr_spmv_dot_partials(n, val.size, 4 if suf == "f32" else 8))
    out = []
    for name, structure in (("lsk_csr_spmv_dot", (P(rd), 0, P(col64), P(vd), P(xd), 0, n - 1)),
                            ("lsk_csr_spmv_c32_dot", (P(off), P(cd), P(vd), P(xd), n))):
        print(f"case {name}_{suf}", flush=True)
        yd = torch.zeros(n, dtype=tdt, device="cuda")
        part = torch.full((slots,), 7.0, dtype=tdt, device="cuda")
        count = ctypes.c_int64(0)
        _lib.call(f"{name}_{suf}", n, val.size, *structure, P(yd), 0, P(w), P(part),
                  slots, ctypes.byref(count), stream())
        assert torch.equal(yd, ref) and count.value == slots
        out.append(part.clone())
    assert torch.equal(out[0], out[1])
    # tuner on the compressed form returns one of its candidates and a time
    remap, ms = ctypes.c_int(-1), ctypes.c_float(0.0)
    yd = torch.zeros(n, dtype=tdt, device="cuda")
    _lib.call(f"lsk_csr_spmv_c32_tune_{suf}", n, val.size, P(off), P(cd), P(vd), P(xd), n, P(yd), stream(),
              ctypes.byref(remap), ctypes.byref(ms))
    assert remap.value in (0, 16, 64, 256, 1024, 4096) and ms.value > 0 and torch.equal(yd, ref)


def test_csr_compress_cols_limits():
    """A window too wide for 32-bit offsets is refused; an empty window marks every column."""
    col = dev(np.array([0, 5, 2**33, -3], dtype=np.int64))
    out = torch.zeros(4, dtype=torch.int32, device="cuda")
    with pytest.raises(_lib.LskError):
        _lib.call("lsk_csr_compress_cols_u32", 4, P(col), 0, 2**32, P(out), stream())
    _lib.call("lsk_csr_compress_cols_u32", 4, P(col), 0, 2**32 - 3, P(out), stream())
    assert out.cpu().numpy().view(np.uint32).tolist() == [0, 5, 0xFFFFFFFF, 0xFFFFFFFF]
    _lib.call("lsk_csr_compress_cols_u32", 4, P(col), 10, 9, P(out), stream())
    assert out.cpu().numpy().view(np.uint32).tolist() == [0xFFFFFFFF] * 4
    # a product against an empty window: y <- 0
    off = dev(np.array([0, 2, 4], dtype=np.int32))
    y = torch.ones(2, dtype=torch.float64, device="cuda")
    val = torch.ones(4, dtype=torch.float64, device="cuda")
    _lib.call("lsk_csr_spmv_c32_f64", 2, 4, P(off), P(out), P(val), None, 0, P(y), 0, stream())
    assert y.tolist() == [0.0, 0.0]
    # rects that cannot become offsets: a gap, a rect outside the piece, too many entries
    o3 = torch.zeros(3, dtype=torch.int32, device="cuda")
    lib = _lib.lib()
    for rects, k_base, nnz in ((np.array([[0, 1], [3, 4]]), 0, 5), (np.array([[0, 1], [2, 9]]), 0, 5),
                               (np.array([[5, 6], [7, 8]]), 6, 4), (np.array([[0, 1], [2, 3]]), 0, 2**32 + 1)):
        rd = dev(rects.astype(np.int64))
        assert lib.lsk_csr_compress_rows_u32(2, nnz, P(rd), k_base, P(o3), stream()) == _lib.LSK_ERR_UNSUPPORTED
    rd = dev(np.array([[6, 6], [7, 6]], dtype=np.int64))  # the second row is empty
    _lib.call("lsk_csr_compress_rows_u32", 2, 4, P(rd), 6, P(o3), stream())
    assert o3.tolist() == [0, 1, 1]


def coo_c32_case(suf, r, c, v, xs_dev, x_lo, x_hi, y0_piece, y_lo, y_hi, want, lim, row_origin):
    """The same product on 32-bit copies of the index fields (rows counted from row_origin)."""
    n = v.size
    rd, cd, vd = dev(r), dev(c), dev(v)
    r32 = torch.empty(n, dtype=torch.int32, device="cuda")
    c32 = torch.empty(n, dtype=torch.int32, device="cuda")
    _lib.call("lsk_csr_compress_cols_u32", n, P(rd), row_origin, row_origin + 2**20, P(r32), stream())
    _lib.call("lsk_csr_compress_cols_u32", n, P(cd), x_lo, x_hi, P(c32), stream())
    yd = dev(y0_piece)
    _lib.call(f"lsk_coo_spmv_c32_{suf}", n, P(r32), P(c32), P(vd), P(xs_dev), x_hi - x_lo + 1, P(yd),
              y_lo - row_origin, y_hi - row_origin, stream())
    err = np.abs(yd.cpu().numpy().astype(np.float64) - want)
    assert np.all(err <= lim), ("32-bit index fields", float(err.max()), int(np.argmax(err - lim)))


@pytest.fixture(params=[1, 0], ids=["wave-scan", "workgroup-chunks"])
def coo_variant(request):
    """Both forms of the COO kernel (option coo.variant); the default is restored afterwards."""
    _lib.call("lsk_set_option", b"coo.variant", request.param)
    yield request.param
    _lib.call("lsk_set_option", b"coo.variant", 1)


@pytest.mark.parametrize("suf", ["f32", "f64"])
def test_coo_spmv_run_shapes(suf, coo_variant):
    """Entry counts around the wave (64), wave-tile (512) and workgroup (2048) boundaries and
    runs of equal rows of every length from 1 to several waves, sorted and unsorted."""
    npdt, tdt, tol = DT[suf]
    rng = np.random.default_rng(27)
    n_rows, n_cols = 300, 257
    x = rng.standard_normal(n_cols).astype(npdt)
    for nnz in (1, 2, 63, 64, 65, 127, 128, 511, 512, 513, 1000, 2047, 2048, 2049, 4096, 10000):
        for shape in ("short-runs", "long-runs", "unsorted"):
            if shape == "short-runs":
                lens = rng.integers(1, 6, size=nnz)
            elif shape == "long-runs":
                lens = rng.integers(40, 700, size=nnz)
            else:
                lens = np.ones(nnz, dtype=np.int64)
            ids = rng.integers(0, n_rows, size=nnz) if shape == "unsorted" else np.sort(rng.integers(0, n_rows, size=nnz))
            r = np.repeat(ids, lens)[:nnz].astype(np.int64)
            c = rng.integers(0, n_cols, size=nnz).astype(np.int64)
            v = rng.standard_normal(nnz).astype(npdt)
            print(f"case coo_spmv_{suf} variant={coo_variant} nnz={nnz} {shape}", flush=True)
            y0 = rng.standard_normal(n_rows).astype(npdt)
            want = y0.copy()
            oracle.coo_matvec(r, c, v, x, 0, n_cols - 1, want, 0, n_rows - 1)
            bound = np.zeros(n_rows, dtype=npdt)
            oracle.coo_matvec(r, c, np.abs(v), np.abs(x), 0, n_cols - 1, bound, 0, n_rows - 1)
            yd, rd, cd, vd, xd = dev(y0), dev(r), dev(c), dev(v), dev(x)
            _lib.call(f"lsk_coo_spmv_{suf}", nnz, P(rd), P(cd), P(vd), P(xd), 0, n_cols - 1,
                      P(yd), 0, n_rows - 1, stream())
            err = np.abs(yd.cpu().numpy().astype(np.float64) - want)
            lim = tol * (bound.astype(np.float64) + np.abs(want) + 1e-300)
            assert np.all(err <= lim), (nnz, shape, float(err.max()), int(np.argmax(err - lim)))
            if coo_variant == 1:
                coo_c32_case(suf, r, c, v, xd, 0, n_cols - 1, y0, 0, n_rows - 1, want, lim, 0)


@pytest.mark.parametrize("suf", ["f32", "f64"])
def test_coo_spmv(suf, coo_variant):
    npdt, tdt, tol = DT[suf]
    rng = np.random.default_rng(26)
    offs, cfs = oracle.benchmark_stencil_offsets(4)
    _, row, col, val = oracle.linearized_stencil((9, 8, 7), offs, cfs, dtype=npdt, coo=True)
    n = 9 * 8 * 7
    x = rng.standard_normal(n).astype(npdt)
    y0 = rng.standard_normal(n).astype(npdt)
    cases = {"row-sorted": np.arange(val.size), "shuffled": rng.permutation(val.size)}
    for name, order in cases.items():
        r, c, v = row[order].copy(), col[order].copy(), val[order].copy()
        for (y_lo, y_hi, x_lo, x_hi) in [(0, n - 1, 0, n - 1), (100, 299, 0, n - 1), (0, n - 1, 50, 400), (7, 7, 0, n - 1)]:
            print(f"case coo_spmv_{suf} {name} y=[{y_lo},{y_hi}] x=[{x_lo},{x_hi}]", flush=True)
            want = np.ascontiguousarray(y0[y_lo:y_hi + 1]).copy()
            xs = np.ascontiguousarray(x[x_lo:x_hi + 1])
            oracle.coo_matvec(r, c, v, xs, x_lo, x_hi, want, y_lo, y_hi)
            bound = np.zeros(y_hi - y_lo + 1, dtype=npdt)
            oracle.coo_matvec(r, c, np.abs(v), np.abs(xs), x_lo, x_hi, bound, y_lo, y_hi)
            yd, rd, cd, vd, xd = dev(y0[y_lo:y_hi + 1]), dev(r), dev(c), dev(v), dev(xs)
            _lib.call(f"lsk_coo_spmv_{suf}", v.size, P(rd), P(cd), P(vd), P(xd), x_lo, x_hi,
                      P(yd), y_lo, y_hi, stream())
            err = np.abs(yd.cpu().numpy().astype(np.float64) - want)
            assert np.all(err <= tol * (bound + np.abs(want) + 1e-300)), (name, err.max())
            if coo_variant == 1:  # rows counted from 0 and from a row inside the matrix
                for origin in (0, min(y_lo, 5)):
                    coo_c32_case(suf, r, c, v, xd, x_lo, x_hi, y0[y_lo:y_hi + 1], y_lo, y_hi, want,
                                 tol * (bound + np.abs(want) + 1e-300), origin)
    # one row with many entries (runs longer than the on-chip cut) and duplicates
    r = np.full(5000, 3, dtype=np.int64)
    c = rng.integers(0, 10, 5000).astype(np.int64)
    v = rng.standard_normal(5000).astype(npdt)
    xs = rng.standard_normal(10).astype(npdt)
    want = np.zeros(8, dtype=npdt)
    oracle.coo_matvec(r, c, v, xs, 0, 9, want, 0, 7)
    yd, rd, cd, vd, xd = torch.zeros(8, dtype=tdt, device="cuda"), dev(r), dev(c), dev(v), dev(xs)
    _lib.call(f"lsk_coo_spmv_{suf}", 5000, P(rd), P(cd), P(vd), P(xd), 0, 9, P(yd), 0, 7, stream())
    scale = float(np.abs(v.astype(np.float64) * xs[c]).sum())
    assert abs(float(yd[3]) - float(want[3])) <= tol * scale * 4
    assert float(yd.abs().sum() - yd[3].abs()) == 0.0


@pytest.mark.parametrize("suf", ["f32", "f64"])
def test_csr_spmv_dot(suf):
    """The fused product + dot: y bit-identical with the plain product, the partial sums add up
    to dot(y, w); regular tiles, batched tiles, irregular rows, empty pieces, beta = 1."""
    npdt, tdt, tol = DT[suf]
    rng = np.random.default_rng(33)
    n_cols = 5000
    cases = {}
    offs, cfs = oracle.benchmark_stencil_offsets(4)
    r, _, c, v = oracle.linearized_stencil((20, 18, 22), offs, cfs, dtype=npdt)
    cases["stencil27"] = (r, c, v, r.shape[0])
    offs, cfs = oracle.benchmark_stencil_offsets(3)
    r, _, c, v = oracle.linearized_stencil((30, 31, 29), offs, cfs, dtype=npdt)
    cases["stencil7"] = (r, c, v, r.shape[0])
    lens = rng.integers(0, 6, size=3000)
    lens[::97] = 3000
    starts = np.concatenate([[0], np.cumsum(lens)])
    cases["short-and-long"] = (np.stack([starts[:-1], starts[1:] - 1], axis=1).astype(np.int64),
                               rng.integers(0, n_cols, size=int(starts[-1])).astype(np.int64),
                               rng.standard_normal(int(starts[-1])).astype(npdt), n_cols)
    lens = rng.integers(0, 40, size=700)
    starts = np.concatenate([[0], np.cumsum(lens)])
    rects = np.stack([starts[:-1], starts[1:] - 1], axis=1).astype(np.int64)
    cases["permuted"] = (np.ascontiguousarray(rects[rng.permutation(700)]),
                         rng.integers(0, n_cols, size=int(starts[-1])).astype(np.int64),
                         rng.standard_normal(int(starts[-1])).astype(npdt), n_cols)
    for name, (rects, col, val, ncol) in cases.items():
        rows = rects.shape[0]
        x = rng.standard_normal(ncol).astype(npdt)
        w = rng.standard_normal(rows).astype(npdt)
        y0 = rng.standard_normal(rows).astype(npdt)
        for beta in (0, 1):
            print(f"case csr_spmv_dot_{suf} {name} rows={rows} beta={beta}", flush=True)
            rd, cd, vd, xd, wd = dev(rects), dev(col), dev(val), dev(x), dev(w)
            y_plain, y_fused = dev(y0), dev(y0)
            _lib.call(f"lsk_csr_spmv_{suf}", rows, val.size, P(rd), 0, P(cd), P(vd), P(xd), 0, ncol - 1,
                      P(y_plain), beta, stream())
            slots = _lib.lib().lsk_csr_spmv_dot_partials(rows, val.size, np.dtype(npdt).itemsize)
            partials = torch.full((slots,), 5.0, dtype=tdt, device="cuda")
            count = ctypes.c_int64(-1)
            _lib.call(f"lsk_csr_spmv_dot_{suf}", rows, val.size, P(rd), 0, P(cd), P(vd), P(xd), 0, ncol - 1,
                      P(y_fused), beta, P(wd), P(partials), slots, ctypes.byref(count), stream())
            assert 0 < count.value <= slots
            assert torch.equal(y_plain, y_fused)
            out = torch.zeros(1, dtype=tdt, device="cuda")
            _lib.call(f"lsk_sum_partials_{suf}", count.value, P(partials), P(out), None, 0, stream())
            yh = y_fused.cpu().numpy().astype(np.float64)
            exact = float(np.sum(yh * w))
            bound = float(np.sum(np.abs(yh * w)))
            assert abs(float(out) - exact) <= (1e-13 if suf == "f64" else 2e-6) * bound + 1e-300
            # one slot too few is refused
            rc = getattr(_lib.lib(), f"lsk_csr_spmv_dot_{suf}")(
                rows, val.size, P(rd), 0, P(cd), P(vd), P(xd), 0, ncol - 1, P(y_fused), beta, P(wd),
                P(partials), count.value - 1, ctypes.byref(count), stream())
            assert rc == -2
    # no rows: nothing to add up
    count = ctypes.c_int64(-1)
    one = torch.zeros(1, dtype=tdt, device="cuda")
    _lib.call(f"lsk_csr_spmv_dot_{suf}", 0, 0, None, 0, None, None, None, 0, -1, None, 0, P(one), P(one), 1,
              ctypes.byref(count), stream())
    assert count.value == 0


@pytest.mark.parametrize("suf", ["f32", "f64"])
def test_csr_spmv_tune(suf):
    """The set-up-time placement tuner: returns one of its candidates with a positive time,
    leaves y = A x behind and the library option as it was."""
    npdt, tdt, tol = DT[suf]
    offs, cfs = oracle.benchmark_stencil_offsets(4)
    rects, _, col, val = oracle.linearized_stencil((40, 40, 40), offs, cfs, dtype=npdt)
    n = rects.shape[0]
    x = np.random.default_rng(30).standard_normal(n).astype(npdt)
    want = np.zeros(n, dtype=npdt)
    oracle.csr_matvec(rects, 0, col, val, x, 0, n - 1, want)
    before = ctypes.c_int(-7)
    _lib.call("lsk_get_option", b"xcd_remap", ctypes.byref(before))
    rd, cd, vd, xd = dev(rects), dev(col), dev(val), dev(x)
    yd = torch.full((n,), 9.0, dtype=tdt, device="cuda")
    remap, ms = ctypes.c_int(-1), ctypes.c_float(-1.0)
    print(f"case csr_spmv_tune_{suf} rows={n}", flush=True)
    _lib.call(f"lsk_csr_spmv_tune_{suf}", n, val.size, P(rd), 0, P(cd), P(vd), P(xd), 0, n - 1, P(yd),
              stream(), ctypes.byref(remap), ctypes.byref(ms))
    assert remap.value in (0, 16, 64, 256, 1024, 4096) and ms.value > 0.0
    after = ctypes.c_int(-7)
    _lib.call("lsk_get_option", b"xcd_remap", ctypes.byref(after))
    assert after.value == before.value
    np.testing.assert_allclose(yd.cpu().numpy(), want, rtol=0, atol=tol * 40)
    # every candidate placement gives the same (bit-identical) result
    results = []
    for c in (0, 16, 64, 256, 1024, 4096, -1):
        _lib.call("lsk_set_option", b"xcd_remap", c)
        y2 = torch.zeros(n, dtype=tdt, device="cuda")
        _lib.call(f"lsk_csr_spmv_{suf}", n, val.size, P(rd), 0, P(cd), P(vd), P(xd), 0, n - 1, P(y2), 0, stream())
        results.append(y2.cpu().numpy())
    _lib.call("lsk_set_option", b"xcd_remap", before.value)
    for r in results[1:]:
        assert np.array_equal(r, results[0])


def test_csr_halo_rows():
    """Rows of a piece that read columns outside an interval, against a direct numpy scan."""
    offs, cfs = oracle.benchmark_stencil_offsets(4)
    rects, _, col, _ = oracle.linearized_stencil((9, 7, 8), offs, cfs)
    n = rects.shape[0]
    rd, cd = dev(rects), dev(col)
    for (r0, r1, lo, hi) in [(0, n - 1, 0, n - 1), (100, 299, 100, 299), (100, 299, 0, n - 1), (100, 299, 150, 160),
                             (0, 55, 0, 55), (n - 56, n - 1, n - 56, n - 1), (200, 200, 0, 10), (10, 30, 500, 600)]:
        piece = np.ascontiguousarray(rects[r0:r1 + 1])
        rows = piece.shape[0]
        k_lo, k_hi = int(piece[0, 0]), int(piece[-1, 1])
        below, above = -1, rows
        for i in range(rows):
            seg = col[piece[i, 0]:piece[i, 1] + 1]
            if (seg < lo).any():
                below = i
            if (seg > hi).any() and above == rows:
                above = i
        out = torch.zeros(2, dtype=torch.int64, device="cuda")
        print(f"case csr_halo_rows rows=[{r0},{r1}] interval=[{lo},{hi}]", flush=True)
        _lib.call("lsk_csr_halo_rows_i64", rows, k_hi - k_lo + 1, P(rd[r0:r1 + 1]), k_lo, P(cd[k_lo:k_hi + 1]),
                  lo, hi, P(out), stream())
        assert out.tolist() == [below, above]
    _lib.call("lsk_csr_halo_rows_i64", 0, 0, None, 0, None, 0, 5, P(out), stream())
    assert out.tolist() == [-1, 0]


def test_rects_to_indptr_bit_exact():
    offs, cfs = oracle.benchmark_stencil_offsets(4)
    rects, _, _, _ = oracle.linearized_stencil((5, 6, 7), offs, cfs)
    for r0, r1 in [(0, rects.shape[0] - 1), (17, 120), (33, 33)]:
        piece = np.ascontiguousarray(rects[r0:r1 + 1])
        out = torch.zeros(piece.shape[0] + 1, dtype=torch.int64, device="cuda")
        pd = dev(piece)
        _lib.call("lsk_rects_to_indptr_i64", piece.shape[0], P(pd), P(out), stream())
        assert np.array_equal(out.cpu().numpy(), oracle.rects_to_indptr(piece))


def test_coo_block_order_and_gather():
    """The stable sort by column block and the gathers, against numpy; then the product on the
    re-ordered fields equals the product on the row-ordered ones up to the order of additions."""
    rng = np.random.default_rng(81)
    for n, extent, block in ((1, 10, 4), (1000, 77, 8), (300_000, 1 << 20, 1 << 12), (70_000, 5000, 5000), (4097, 3, 1)):
        print(f"case coo_block_order n={n} extent={extent} block={block}", flush=True)
        c = rng.integers(0, extent, size=n).astype(np.uint32)
        c[rng.integers(0, n, size=max(n // 50, 1))] = 0xFFFFFFFF  # columns outside the window
        cd = dev(c.view(np.int32))
        order = torch.full((n,), -1, dtype=torch.int32, device="cuda")
        _lib.call("lsk_coo_block_order_u32", n, P(cd), extent, block, P(order), stream())
        last = -(-extent // block)
        keys = np.where(c == 0xFFFFFFFF, last, c // block)
        want = np.argsort(keys, kind="stable")
        got = order.cpu().numpy().view(np.uint32)
        assert np.array_equal(got, want.astype(np.uint32))
        for arr, name in ((c.view(np.int32), "lsk_gather_u32"), (rng.standard_normal(n).astype(np.float32), "lsk_gather_f32"),
                          (rng.standard_normal(n), "lsk_gather_f64")):
            src = dev(arr)
            dst = torch.empty_like(src)
            _lib.call(name, n, P(src), P(order), P(dst), stream())
            assert np.array_equal(dst.cpu().numpy(), arr[want])
    # product on re-ordered fields
    n_rows, n_cols, nnz = 5000, 40_000, 200_000
    r = np.sort(rng.integers(0, n_rows, size=nnz)).astype(np.uint32)
    c = rng.integers(0, n_cols, size=nnz).astype(np.uint32)
    v = rng.standard_normal(nnz)
    x = rng.standard_normal(n_cols)
    rd, cd, vd, xd = dev(r.view(np.int32)), dev(c.view(np.int32)), dev(v), dev(x)
    y0 = torch.zeros(n_rows, dtype=torch.float64, device="cuda")
    _lib.call("lsk_coo_spmv_c32_f64", nnz, P(rd), P(cd), P(vd), P(xd), n_cols, P(y0), 0, n_rows - 1, stream())
    order = torch.empty(nnz, dtype=torch.int32, device="cuda")
    _lib.call("lsk_coo_block_order_u32", nnz, P(cd), n_cols, 4096, P(order), stream())
    rb, cb, vb = torch.empty_like(rd), torch.empty_like(cd), torch.empty_like(vd)
    _lib.call("lsk_gather_u32", nnz, P(rd), P(order), P(rb), stream())
    _lib.call("lsk_gather_u32", nnz, P(cd), P(order), P(cb), stream())
    _lib.call("lsk_gather_f64", nnz, P(vd), P(order), P(vb), stream())
    y1 = torch.zeros(n_rows, dtype=torch.float64, device="cuda")
    _lib.call("lsk_coo_spmv_c32_f64", nnz, P(rb), P(cb), P(vb), P(xd), n_cols, P(y1), 0, n_rows - 1, stream())
    bound = np.zeros(n_rows)
    np.add.at(bound, r.astype(np.int64), np.abs(v * x[c.astype(np.int64)]))
    assert np.all(np.abs((y1 - y0).cpu().numpy()) <= 1e-13 * (bound + 1e-300))


def test_inclusive_sum_bit_exact():
    rng = np.random.default_rng(71)
    for n in (1, 2, 255, 256, 257, 100_003, 3_000_000):
        print(f"case inclusive_sum n={n}", flush=True)
        a = rng.integers(0, 40, size=n).astype(np.int64)
        src, out = dev(a), torch.zeros(n, dtype=torch.int64, device="cuda")
        _lib.call("lsk_inclusive_sum_i64", n, P(src), P(out), stream())
        assert np.array_equal(out.cpu().numpy(), np.cumsum(a))
        _lib.call("lsk_inclusive_sum_i64", n, P(src), P(src), stream())  # in place
        assert np.array_equal(src.cpu().numpy(), np.cumsum(a))
    _lib.call("lsk_inclusive_sum_i64", 0, None, None, stream())


def test_index_bounds():
    rng = np.random.default_rng(27)
    for n in (1, 100, 100000, (1 << 21) + 7):
        idx = rng.integers(-5000, 1 << 40, size=n).astype(np.int64)
        out = torch.zeros(2, dtype=torch.int64, device="cuda")
        idx_d = dev(idx)
        _lib.call("lsk_index_bounds_i64", n, P(idx_d), P(out), stream())
        assert out.tolist() == [int(idx.min()), int(idx.max())]
    out = torch.zeros(2, dtype=torch.int64, device="cuda")
    _lib.call("lsk_index_bounds_i64", 0, None, P(out), stream())
    assert out.tolist() == [np.iinfo(np.int64).max, np.iinfo(np.int64).min]


@pytest.mark.parametrize("suf", ["f32", "f64"])
@pytest.mark.parametrize("flag,extent", [(1, (50,)), (2, (9, 13)), (3, (5, 6, 7)), (4, (6, 5, 7)), (4, (1, 1, 4))])
def test_device_stencil_generator_bit_exact(gpu_ctx, suf, flag, extent):
    """The device generators write exactly what the oracle's restatement of
    FillLinearizedCSRStencilTask / FillLinearizedCOOStencilTask writes, for 1 and 3 pieces."""
    from legionsolvers_amd import IndexPartition, IndexSpace
    from legionsolvers_amd import stencil_generator as sg
    npdt, tdt, _ = DT[suf]
    dim, offs, cfs = sg.benchmark_stencil(flag)
    assert (offs, cfs) == oracle.benchmark_stencil_offsets(flag)
    rects, row, col, val = oracle.linearized_stencil(extent, offs, cfs, dtype=npdt, coo=True)
    n = rects.shape[0]
    for pieces in (1, 3):
        ip = IndexPartition.equal(IndexSpace(0, n - 1), pieces)
        m = sg.create_linearized_csr_stencil_matrix(gpu_ctx, extent, offs, cfs, ip, tdt)
        assert m.k_base == 0 and m.nnz == val.size and m.rows == n
        assert np.array_equal(m.rects.cpu().numpy(), rects)
        assert np.array_equal(m.col.cpu().numpy(), col)
        assert np.array_equal(m.val.cpu().numpy().view(np.uint8), val.view(np.uint8))
        c = sg.create_linearized_coo_stencil_matrix(gpu_ctx, extent, offs, cfs, ip, tdt)
        assert np.array_equal(c.row.cpu().numpy(), row) and np.array_equal(c.col.cpu().numpy(), col)
        assert np.array_equal(c.val.cpu().numpy().view(np.uint8), val.view(np.uint8))
        kp = m.create_kernel_partition_from_range_partition(ip)
        gp = m.create_domain_partition_from_kernel_partition(IndexSpace(0, n - 1), kp)
        from oracle import partition as part
        for i, piece in enumerate(ip.pieces):
            want_k = part.image_range(rects, (piece.lo[0], piece.hi[0]))
            assert kp.intervals[i] == part.bounds(want_k)
            assert gp.intervals[i] == part.bounds(part.image(col, want_k, (0, n - 1)))
        kc = c.create_kernel_partition_from_range_partition(ip)
        assert kc.intervals == kp.intervals  # preimage of rows == image-range of rects here


# ------------------------------------------------------------------ sliced row patterns -----

def sliced_padding_too_costly(suf, rects, nnz):
    """The library's rule for declining: entries padded to the block's longest row (rows the last
    block lacks left out) * sizeof(T) > nnz * (sizeof(T) + 3)."""
    lens = np.maximum(rects[:, 1] - rects[:, 0] + 1, 0)
    br, esz = BLOCK_ROWS[suf], np.dtype(DT[suf][0]).itemsize
    padded = sum(int(lens[s:s + br].size) * int(lens[s:s + br].max()) for s in range(0, rects.shape[0], br))
    return padded * esz > nnz * (esz + 3)


def sliced_expected_padding(rects, k_base, col, row_first, br):
    """Entries of the sliced copy: every block of br rows padded to the list it runs on -- its longest
    row's, when every row's list of distances is an ordered part of that; else the shortest list of the
    matrix that all of them are ordered parts of (else, again, the longest row's: a block without a
    common list)."""
    def part_of(a, b):
        it = iter(b)
        return all(d in it for d in a)

    rows = rects.shape[0]
    lists = [tuple((col[rects[r, 0] - k_base:rects[r, 1] - k_base + 1] - (row_first + r)).tolist()) for r in range(rows)]
    table = sorted(set(lists), key=len)
    total = 0
    for s0 in range(0, rows, br):
        here = set(lists[s0:s0 + br])
        longest = max(here, key=len)
        if all(part_of(l, longest) for l in here):
            total += br * len(longest)
            continue
        other = next((t for t in table if all(part_of(l, t) for l in here)), longest)
        total += br * len(other)
    return total


def sliced_case(suf, rects, k_base, col, val, x, x_lo, x_hi, row_first, tag, expect=True, uniform=None):
    """lsk_csr_sliced_create_* + lsk_csr_sliced_spmv_* on a piece: bit-exact against the oracle's
    ascending-k loop (separate multiply and add), for beta = 0 and 1 and for a range of rows."""
    npdt, tdt, _ = DT[suf]
    rows = rects.shape[0]
    print(f"case csr_sliced_{suf} {tag} rows={rows} nnz={val.size} k_base={k_base} x=[{x_lo},{x_hi}] "
          f"row_first={row_first}", flush=True)
    rd, cd, vd, xd = dev(rects), dev(col), dev(val), dev(x)
    handle = ctypes.c_void_p(0)
    code = getattr(_lib.lib(), f"lsk_csr_sliced_create_{suf}")(rows, val.size, P(rd), k_base, P(cd), P(vd), row_first,
                                                             x_lo, x_hi, stream(), ctypes.byref(handle))
    if expect and sliced_padding_too_costly(suf, rects, val.size):
        expect = False  # short rows padded to their block's longest would cost more than the columns save
    if not expect:
        assert code == _lib.LSK_ERR_UNSUPPORTED and not handle.value, f"{tag}: create said {code}"
        return None
    assert code == _lib.LSK_OK, f"{tag}: create said {code}: {_lib.lib().lsk_last_error().decode()}"
    info = (ctypes.c_int64 * 9)()
    _lib.call("lsk_csr_sliced_info", handle, info)
    br = BLOCK_ROWS[suf]
    padded = sliced_expected_padding(rects, k_base, col, row_first, br)
    assert info[0] == padded and info[1] == -(-rows // br) and info[5] == br, f"{tag}: info {list(info)}"
    if uniform is not None:
        # (only structures whose blocks all run on one staged list are judged: info[7] says so)
        assert _lib.lib().lsk_csr_sliced_uniform_values(handle) == (uniform if info[7] else 0), f"{tag}: uniform values"
        print(f"     uniform values: {_lib.lib().lsk_csr_sliced_uniform_values(handle)} (blocks without one list: {info[2]})", flush=True)
    try:
        rng = np.random.default_rng(rows)
        y0 = rng.standard_normal(rows).astype(npdt)
        for beta, (ra, rb) in ((0, (0, rows)), (1, (0, rows)), (0, (rows // 3, rows - rows // 5)), (1, (1, min(rows, 66))),
                               (0, (min(rows, br - 3), min(rows, 2 * br + 5))), (1, (rows // 2, rows // 2))):
            want = y0.copy()
            part = want[ra:rb].copy()
            oracle.csr_matvec(rects[ra:rb], k_base, col, val, x, x_lo, x_hi, part, beta=beta)
            want[ra:rb] = part
            yd = dev(y0)
            _lib.call(f"lsk_csr_sliced_spmv_{suf}", handle, ra, rb, P(xd), P(yd), beta, stream())
            got = yd.cpu().numpy()
            assert np.array_equal(got.view(np.uint8), want.view(np.uint8)), \
                f"{tag}: beta={beta} rows {ra}..{rb}: {int((got != want).sum())} rows differ"
        # the fused product + dot: the same y bit for bit, and shares that add up to y . w
        w = rng.standard_normal(rows).astype(npdt)
        want = np.zeros(rows, dtype=npdt)
        oracle.csr_matvec(rects, k_base, col, val, x, x_lo, x_hi, want, beta=0)
        slots = -(-rows // br)
        assert _lib.lib().lsk_csr_sliced_dot_partials(handle, 0, rows) == slots
        if rows > br:
            assert _lib.lib().lsk_csr_sliced_dot_partials(handle, br - 1, br + 1) == 2
        yd, wd = dev(y0), dev(w)
        partials = torch.full((slots + 3,), 7.0, dtype=tdt, device="cuda")
        count = ctypes.c_int64(-1)
        _lib.call(f"lsk_csr_sliced_spmv_dot_{suf}", handle, 0, rows, P(xd), P(yd), 0, P(wd), P(partials), slots + 3,
                  ctypes.byref(count), stream())
        assert count.value == slots and np.array_equal(yd.cpu().numpy().view(np.uint8), want.view(np.uint8)), f"{tag}: fused"
        got = partials.cpu().numpy()
        assert np.all(got[slots:] == 7.0), f"{tag}: fused form wrote past its shares"
        exact = float(np.dot(want.astype(np.float64), w.astype(np.float64)))
        scale = float(np.dot(np.abs(want).astype(np.float64), np.abs(w).astype(np.float64))) + 1e-300
        assert abs(float(got[:slots].astype(np.float64).sum()) - exact) <= (1e-13 if suf == "f64" else 1e-5) * scale, f"{tag}: fused dot"
        code = getattr(_lib.lib(), f"lsk_csr_sliced_spmv_dot_{suf}")(handle, 0, rows, P(xd), P(yd), 0, P(wd), P(partials),
                                                                   slots - 1, ctypes.byref(count), stream())
        assert code == _lib.LSK_ERR_ARG and count.value == 0, f"{tag}: short partials buffer accepted"
        # w the very rows of x (CG's p . A p: the kernel then takes w from its staged windows of x): the same
        # shares, bit for bit, as with a copy of those rows somewhere else in memory
        shift = row_first - x_lo
        if shift >= 0 and shift + rows <= x.size:
            w_view = xd[shift:shift + rows]
            w_copy = w_view.clone()
            got = []
            for wv in (w_view, w_copy):
                yd = dev(y0)
                part = torch.full((slots,), 7.0, dtype=tdt, device="cuda")
                _lib.call(f"lsk_csr_sliced_spmv_dot_{suf}", handle, 0, rows, P(xd), P(yd), 0, P(wv), P(part), slots,
                          ctypes.byref(count), stream())
                assert np.array_equal(yd.cpu().numpy().view(np.uint8), want.view(np.uint8)), f"{tag}: fused, w = x"
                got.append(part)
            assert torch.equal(got[0], got[1]), f"{tag}: shares differ when w is the rows of x"
            if info[7]:
                xb2 = dev(rng.standard_normal(x.size).astype(npdt))
                outs = []
                for wa2, wb2 in ((w_view, xb2[shift:shift + rows]), (w_copy, xb2[shift:shift + rows].clone())):
                    ya, yb = dev(y0), dev(y0)
                    pa = torch.full((slots,), 7.0, dtype=tdt, device="cuda")
                    pb = torch.full((slots,), 7.0, dtype=tdt, device="cuda")
                    _lib.call(f"lsk_csr_sliced_spmv2_dot_{suf}", handle, 0, rows, P(xd), P(xb2), P(ya), P(yb), 0, P(wa2),
                              P(wb2), P(pa), P(pb), slots, ctypes.byref(count), stream())
                    outs.append((pa, pb))
                assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1]), \
                    f"{tag}: the pair's shares differ when w is the rows of x"
                assert torch.equal(outs[0][0], got[0]), f"{tag}: the pair's first shares are not the single product's"
        sliced_pair_case(suf, handle, rows, br, xd, x.size, y0, wd, partials[:slots].clone(), tag,
                         must_serve=bool(info[7]))
        assert info[7] == 1 or not tag.startswith("stencil"), f"{tag}: the two-vector product should serve a stencil"
    finally:
        _lib.call("lsk_csr_sliced_destroy", handle)
    return list(info)


def sliced_pair_case(suf, handle, rows, br, xd, x_size, y0, wd, shares_a, tag, must_serve):
    """lsk_csr_sliced_spmv2_*: two right-hand sides with one pass over the values; each result, and
    each set of dot shares, bit for bit what the single product gives (shares_a: those of xd with wd)."""
    npdt, tdt, _ = DT[suf]
    lib = _lib.lib()
    rng = np.random.default_rng(rows + 1)
    xb = dev(rng.standard_normal(x_size).astype(npdt))
    wb = dev(rng.standard_normal(rows).astype(npdt))
    slots = -(-rows // br)
    for beta, (ra, rb) in ((0, (0, rows)), (1, (0, rows)), (0, (rows // 3, rows - rows // 5)),
                           (1, (min(rows, br - 3), min(rows, 2 * br + 5)))):
        want = []
        for xv in (xd, xb):
            yd = dev(y0)
            _lib.call(f"lsk_csr_sliced_spmv_{suf}", handle, ra, rb, P(xv), P(yd), beta, stream())
            want.append(yd)
        ya, yb = dev(y0), dev(y0)
        code = getattr(lib, f"lsk_csr_sliced_spmv2_{suf}")(handle, ra, rb, P(xd), P(xb), P(ya), P(yb), beta, stream())
        if not must_serve:   # the info call said so: the product declines, whatever the rows
            assert code == _lib.LSK_ERR_UNSUPPORTED, f"{tag}: spmv2 said {code} for a structure it does not serve"
            print(f"case csr_sliced_spmv2_{suf} {tag}: declined (blocks without one staged list)", flush=True)
            return
        assert code == _lib.LSK_OK, f"{tag}: spmv2 said {code}: {lib.lsk_last_error().decode()}"
        assert torch.equal(ya, want[0]) and torch.equal(yb, want[1]), f"{tag}: spmv2 beta={beta} rows {ra}..{rb}"
    # the fused form: the same results, and each system's shares those of its single product
    yd = dev(y0)
    pb_single = torch.zeros(slots, dtype=tdt, device="cuda")
    count = ctypes.c_int64(-1)
    _lib.call(f"lsk_csr_sliced_spmv_dot_{suf}", handle, 0, rows, P(xb), P(yd), 0, P(wb), P(pb_single), slots,
              ctypes.byref(count), stream())
    ya, yb = dev(y0), dev(y0)
    pa = torch.full((slots + 2,), 7.0, dtype=tdt, device="cuda")
    pb = torch.full((slots + 2,), 7.0, dtype=tdt, device="cuda")
    _lib.call(f"lsk_csr_sliced_spmv2_dot_{suf}", handle, 0, rows, P(xd), P(xb), P(ya), P(yb), 0, P(wd), P(wb), P(pa), P(pb),
              slots + 2, ctypes.byref(count), stream())
    assert count.value == slots
    assert torch.equal(yb, yd) and torch.equal(pa[:slots], shares_a) and torch.equal(pb[:slots], pb_single), f"{tag}: spmv2_dot"
    assert bool((pa[slots:] == 7.0).all()) and bool((pb[slots:] == 7.0).all()), f"{tag}: spmv2_dot wrote past its shares"
    assert getattr(lib, f"lsk_csr_sliced_spmv2_{suf}")(handle, 0, rows, P(xd), P(xb), P(ya), P(ya), 0, stream()) == _lib.LSK_ERR_ARG
    # the pair's own placement tuner: a candidate and its time come back, the option it tries is restored,
    # and the vectors hold the two products
    held, remap, ms = ctypes.c_int(-1), ctypes.c_int(-1), ctypes.c_float(-1.0)
    _lib.call("lsk_get_option", b"xcd_remap", ctypes.byref(held))
    ya, yb = dev(y0), dev(y0)
    _lib.call(f"lsk_csr_sliced_tune2_{suf}", handle, P(xd), P(xb), P(ya), P(yb), stream(), ctypes.byref(remap), ctypes.byref(ms))
    now = ctypes.c_int(-2)
    _lib.call("lsk_get_option", b"xcd_remap", ctypes.byref(now))
    assert remap.value in (0, 2, 4, 16, 64, 256, 1024, 4096) and ms.value > 0.0 and now.value == held.value, f"{tag}: tune2"
    for got, xv in ((ya, xd), (yb, xb)):
        yd = dev(y0)
        _lib.call(f"lsk_csr_sliced_spmv_{suf}", handle, 0, rows, P(xv), P(yd), 0, stream())
        assert torch.equal(got, yd), f"{tag}: tune2 left something else than the products"
    assert getattr(lib, f"lsk_csr_sliced_tune2_{suf}")(handle, P(xd), None, P(ya), P(yb), stream(), ctypes.byref(remap),
                                                       ctypes.byref(ms)) == _lib.LSK_ERR_ARG


@pytest.fixture(params=[(1, -1, 1, 1), (0, 1, 1, 0), (1, 0, 1, 0), (1, 1, 0, 0), (0, 0, 1, 1)],
                ids=["the-defaults", "block-staged,pair-by-waves-through-registers", "wave-staged-through-registers,pair-block-staged",
                     "wave-staged,entry-by-entry", "block-staged,pair-block-staged"])
def sliced_forms(request):
    """The forms of the sliced-pattern product (option sliced.form; sliced.runs: lists of runs of three
    taken run by run or entry by entry; sliced.dma: windows of x from memory straight into LDS) and of
    its two-vector variant (sliced.pair_form); the defaults are restored afterwards."""
    form, pair_form, runs, dma = request.param
    defaults = {}
    for key, value in ((b"sliced.form", form), (b"sliced.pair_form", pair_form), (b"sliced.runs", runs),
                       (b"sliced.dma", dma)):
        old = ctypes.c_int(0)
        _lib.call("lsk_get_option", key, ctypes.byref(old))
        defaults[key] = old.value
        _lib.call("lsk_set_option", key, value)
    yield request.param
    for key, value in defaults.items():
        _lib.call("lsk_set_option", key, value)


@pytest.mark.parametrize("suf", ["f64", "f32"])
def test_csr_sliced_patterns_bit_exact(suf, sliced_forms):
    npdt = DT[suf][0]
    rng = np.random.default_rng(17)
    # (5, 9, 300), (4, 3, 1100), (3, 5, 130): grids whose planes do not end where blocks of 512 or 1024 rows
    # end, so that some block holds the last line of one plane and the first of the next -- two lists of
    # equal length, neither an ordered part of the other; such a block runs on the interior row's list
    for flag, extent in ((1, (1000,)), (2, (37, 41)), (3, (9, 11, 13)), (4, (7, 12, 70)), (4, (3, 3, 3)), (4, (5, 9, 300)),
                         (4, (4, 3, 1100)), (3, (3, 5, 130))):
        offs, cfs = oracle.benchmark_stencil_offsets(flag)
        rects, _, col, val = oracle.linearized_stencil(extent, offs, cfs, dtype=npdt, coo=True)
        n = int(np.prod(extent))
        val = rng.standard_normal(val.size).astype(npdt)  # no two entries alike
        x = rng.standard_normal(n).astype(npdt)
        if n < 64:  # one block, padded from 343 entries to a block of 27-entry rows: not worth it
            sliced_case(suf, rects, 0, col, val, x, 0, n - 1, 0, f"stencil{flag}{extent}", expect=False)
            continue
        info = sliced_case(suf, rects, 0, col, val, x, 0, n - 1, 0, f"stencil{flag}{extent}", uniform=0)
        assert info[3] <= 27
        assert info[2] == 0, f"stencil{flag}{extent}: {info[2]} blocks without a common list"
        # every list that serves a block of the 27-point (and of the 1-D 3-point) matrix is runs of three
        # (a row at a line's end has a shorter list of its own, but its block runs on the line's list with
        # a mask); the 5- and 7-point lists have single distances beside their one run
        assert info[8] == (1 if flag in (1, 4) else 0), (flag, extent, info)
        # a piece in the middle: rows a..b of the matrix, entries from k_base, x window = its ghost interval
        a, b = n // 3, n - n // 4
        k_lo, k_hi = int(rects[a, 0]), int(rects[b - 1, 1])
        pc, pv = col[k_lo:k_hi + 1], val[k_lo:k_hi + 1]
        x_lo, x_hi = int(pc.min()), int(pc.max())
        sliced_case(suf, rects[a:b], k_lo, pc, pv, x[x_lo:x_hi + 1], x_lo, x_hi, a, f"stencil{flag}{extent} piece")
        # ... and with a window that misses a column: no structure
        if x_hi > x_lo + 1:
            sliced_case(suf, rects[a:b], k_lo, pc, pv, x[x_lo + 1:x_hi + 1], x_lo + 1, x_hi, a,
                        f"stencil{flag}{extent} short window", expect=False)


@pytest.mark.parametrize("suf", ["f64", "f32"])
def test_csr_sliced_uniform_values(suf):
    """Constant-coefficient stencils (every matrix of the reference's benchmark): each list position has
    one value, the structure says so, and with option sliced.uniform = 1 every form of the product -- one
    vector, ranges of rows, accumulating, with dot shares, two vectors -- runs without the value array and
    still gives the oracle's bits.  Values rewritten in place are judged anew."""
    npdt, tdt, _ = DT[suf]
    lib = _lib.lib()
    rng = np.random.default_rng(23)
    _lib.call("lsk_set_option", b"sliced.uniform", 1)
    try:
        for flag, extent in ((1, (1000,)), (2, (37, 41)), (3, (9, 11, 13)), (4, (7, 12, 70)), (4, (3, 4, 512)), (4, (5, 9, 300))):
            offs, cfs = oracle.benchmark_stencil_offsets(flag)
            rects, _, col, val = oracle.linearized_stencil(extent, offs, cfs, dtype=npdt, coo=True)
            n = int(np.prod(extent))
            x = rng.standard_normal(n).astype(npdt)
            sliced_case(suf, rects, 0, col, val, x, 0, n - 1, 0, f"stencil{flag}{extent} its own coefficients", uniform=1)
            a, b = n // 3, n - n // 4
            k_lo, k_hi = int(rects[a, 0]), int(rects[b - 1, 1])
            pc, pv = col[k_lo:k_hi + 1], val[k_lo:k_hi + 1]
            x_lo, x_hi = int(pc.min()), int(pc.max())
            sliced_case(suf, rects[a:b], k_lo, pc, pv, x[x_lo:x_hi + 1], x_lo, x_hi, a,
                        f"stencil{flag}{extent} piece, its own coefficients", uniform=1)
        # values rewritten in place: one entry off -> not uniform (the product reads the array again);
        # put back -> uniform again; each time the oracle's bits
        offs, cfs = oracle.benchmark_stencil_offsets(4)
        rects, _, col, val = oracle.linearized_stencil((7, 12, 70), offs, cfs, dtype=npdt, coo=True)
        rows = rects.shape[0]
        x = rng.standard_normal(rows).astype(npdt)
        rd, cd, vd, xd = dev(rects), dev(col), dev(val), dev(x)
        handle = ctypes.c_void_p(0)
        _lib.call(f"lsk_csr_sliced_create_{suf}", rows, val.size, P(rd), 0, P(cd), P(vd), 0, 0, rows - 1, stream(),
                  ctypes.byref(handle))
        try:
            assert lib.lsk_csr_sliced_uniform_values(handle) == 1
            for k, expect in ((val.size // 2, 0), (None, 1), (0, 0), (val.size - 1, 0), (None, 1)):
                now = val.copy()
                if k is not None:
                    now[k] = np.nextafter(now[k], npdt(10))   # one bit off
                vd.copy_(torch.from_numpy(now))
                _lib.call(f"lsk_csr_sliced_update_values_{suf}", handle, rows, val.size, P(rd), 0, P(vd), stream())
                want = np.zeros(rows, dtype=npdt)
                oracle.csr_matvec(rects, 0, col, now, x, 0, rows - 1, want, beta=0)
                # a product right behind the update (the verdict on the new values may not be in yet), the
                # verdict, and a product after it
                for ask in (False, True):
                    if ask:
                        assert lib.lsk_csr_sliced_uniform_values(handle) == expect, f"entry {k} changed"
                    yd = torch.empty(rows, dtype=tdt, device="cuda")
                    _lib.call(f"lsk_csr_sliced_spmv_{suf}", handle, 0, rows, P(xd), P(yd), 0, stream())
                    assert np.array_equal(yd.cpu().numpy().view(np.uint8), want.view(np.uint8)), f"entry {k} changed: product"
        finally:
            _lib.call("lsk_csr_sliced_destroy", handle)
    finally:
        _lib.call("lsk_set_option", b"sliced.uniform", 0)


def test_csr_sliced_patterns_decline():
    """Matrices without a few repeating row patterns, rows longer than a pattern may be, rects outside
    the piece: LSK_ERR_UNSUPPORTED, no structure, no abort."""
    rng = np.random.default_rng(5)
    rows, per = 4000, 6
    col = np.sort(rng.integers(0, rows, (rows, per)), axis=1).reshape(-1).astype(np.int64)
    lo = np.arange(rows, dtype=np.int64) * per
    rects = np.stack([lo, lo + per - 1], axis=1)
    val = rng.standard_normal(rows * per)
    x = rng.standard_normal(rows)
    sliced_case("f64", rects, 0, col, val, x, 0, rows - 1, 0, "random columns", expect=False)
    # rows of 40 entries (a band): one pattern, but longer than the table's rows
    rows, per = 300, 40
    lo = np.arange(rows, dtype=np.int64) * per
    rects = np.stack([lo, lo + per - 1], axis=1)
    col = (np.arange(rows)[:, None] + np.arange(per)[None, :]).reshape(-1).astype(np.int64)
    sliced_case("f64", rects, 0, col, rng.standard_normal(rows * per), rng.standard_normal(rows + per), 0,
                rows + per - 1, 0, "long rows", expect=False)
    # heavy padding: one long row per slice among empty ones
    rows = 640
    lens = np.where(np.arange(rows) % 64 == 0, 20, 0)
    hi = np.cumsum(lens) - 1
    rects = np.stack([hi - lens + 1, hi], axis=1).astype(np.int64)
    col = np.concatenate([r + np.arange(20) for r in range(0, rows, 64)]).astype(np.int64)
    sliced_case("f64", rects, 0, col, rng.standard_normal(col.size), rng.standard_normal(rows + 20), 0, rows + 19, 0,
                "mostly padding", expect=False)
    # a rect that points outside the piece's entries
    rects = np.array([[0, 2], [3, 9]], dtype=np.int64)
    sliced_case("f64", rects, 0, np.arange(6, dtype=np.int64), rng.standard_normal(6), rng.standard_normal(6), 0, 5, 0,
                "rect outside", expect=False)


def test_csr_sliced_mixed_and_gapped_rows():
    """Rows with gaps between their rects and empty rows (patterns of length 0) share slices with
    full rows: the mixed-pattern path."""
    rng = np.random.default_rng(8)
    offs, cfs = oracle.benchmark_stencil_offsets(2)
    rects, _, col, val = oracle.linearized_stencil((30, 9), offs, cfs, dtype=np.float64, coo=True)
    rects = rects.copy()
    rects[::7, 1] = rects[::7, 0] - 1        # every 7th row empty
    rects[1::5, 1] -= 1                      # every 5th row ignores its last entry
    n = 270
    info = sliced_case("f64", rects, 0, col, rng.standard_normal(val.size), rng.standard_normal(n), 0, n - 1, 0,
                       "gapped rects")
    assert info[2] == info[1] or info[4] >= 1  # every block mixes patterns: mixed, or masked waves


@pytest.mark.parametrize("suf", ["f64", "f32"])
def test_csr_sliced_update_values_bit_exact(suf):
    """create -> overwrite val in place -> product still gives the OLD values' result (the structure
    holds a copy) -> lsk_csr_sliced_update_values_* -> product equals the oracle on the NEW values,
    bit for bit; masked slices (grid-line ends) and a piece with k_base > 0 included."""
    npdt, tdt, _ = DT[suf]
    rng = np.random.default_rng(29)
    lib = _lib.lib()
    for flag, extent, piece in ((4, (7, 12, 70), False), (3, (9, 11, 13), True), (2, (37, 41), True)):
        offs, cfs = oracle.benchmark_stencil_offsets(flag)
        rects, _, col, val = oracle.linearized_stencil(extent, offs, cfs, dtype=npdt, coo=True)
        n = int(np.prod(extent))
        a, b = (n // 3, n - n // 4) if piece else (0, n)
        k_lo, k_hi = int(rects[a, 0]), int(rects[b - 1, 1])
        pr, pc = rects[a:b], col[k_lo:k_hi + 1]
        x_lo, x_hi = int(pc.min()), int(pc.max())
        rows, nnz = b - a, pc.size
        print(f"case csr_sliced_update_{suf} stencil{flag}{extent} rows={rows} nnz={nnz} k_base={k_lo}", flush=True)
        old = rng.standard_normal(nnz).astype(npdt)
        new = rng.standard_normal(nnz).astype(npdt)
        x = rng.standard_normal(x_hi - x_lo + 1).astype(npdt)
        rd, cd, vd, xd = dev(pr), dev(pc), dev(old), dev(x)
        handle = ctypes.c_void_p(0)
        _lib.call(f"lsk_csr_sliced_create_{suf}", rows, nnz, P(rd), k_lo, P(cd), P(vd), a, x_lo, x_hi, stream(),
                  ctypes.byref(handle))
        try:
            def product():
                yd = torch.full((rows,), 3.0, dtype=tdt, device="cuda")
                _lib.call(f"lsk_csr_sliced_spmv_{suf}", handle, 0, rows, P(xd), P(yd), 0, stream())
                return yd.cpu().numpy()

            def expected(values):
                want = np.zeros(rows, dtype=npdt)
                oracle.csr_matvec(pr, k_lo, pc, values, x, x_lo, x_hi, want, beta=0)
                return want

            assert np.array_equal(product().view(np.uint8), expected(old).view(np.uint8))
            vd.copy_(torch.from_numpy(new))  # the caller rewrites its entries in place
            assert np.array_equal(product().view(np.uint8), expected(old).view(np.uint8)), "the copy followed val by itself?"
            _lib.call(f"lsk_csr_sliced_update_values_{suf}", handle, rows, nnz, P(rd), k_lo, P(vd), stream())
            assert np.array_equal(product().view(np.uint8), expected(new).view(np.uint8))
            # a call that does not describe the piece the structure was made from is refused
            fn = getattr(lib, f"lsk_csr_sliced_update_values_{suf}")
            assert fn(handle, rows - 1, nnz, P(rd), k_lo, P(vd), stream()) == _lib.LSK_ERR_ARG
            assert fn(handle, rows, nnz, P(rd), k_lo + 1, P(vd), stream()) == _lib.LSK_ERR_ARG
            assert fn(handle, rows, nnz, P(rd), k_lo, None, stream()) == _lib.LSK_ERR_ARG
            other = "f32" if suf == "f64" else "f64"
            assert getattr(lib, f"lsk_csr_sliced_update_values_{other}")(handle, rows, nnz, P(rd), k_lo, P(vd),
                                                                          stream()) == _lib.LSK_ERR_ARG
            assert np.array_equal(product().view(np.uint8), expected(new).view(np.uint8))
        finally:
            _lib.call("lsk_csr_sliced_destroy", handle)
