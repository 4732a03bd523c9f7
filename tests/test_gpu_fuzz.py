"""Randomised parity sweep of the SpMV kernels against the CPU oracle: many small matrices with
row-length distributions chosen to hit every path of the wave-tile kernel (1..256 rows per wave,
one or several rects per lane, tiles taken whole / in batches / row by row, pieces that start in
the middle of larger arrays, input windows narrower than the columns), both entry types, both
values of beta, and the same matrices as row-sorted and shuffled COO.  Seeds are fixed."""
import ctypes

import numpy as np
import pytest
import torch

import oracle
from legionsolvers_amd import _lib

pytestmark = pytest.mark.gpu

DT = {"f32": (np.float32, torch.float32, 1e-5), "f64": (np.float64, torch.float64, 1e-13)}


def P(t):
    return ctypes.c_void_p(t.data_ptr() if t is not None else 0)


def stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def row_lengths(rng, kind, rows):
    if kind == "tiny":
        return rng.integers(0, 4, size=rows)
    if kind == "stencil-like":
        return np.full(rows, int(rng.integers(3, 30)))
    if kind == "uniform":
        return rng.integers(0, int(rng.integers(2, 120)), size=rows)
    if kind == "heavy-tail":
        lens = (rng.pareto(1.2, size=rows) * 3).astype(np.int64)
        return np.minimum(lens, 5000)
    if kind == "long":
        return rng.integers(600, 2500, size=rows)
    if kind == "power-law":   # mean ~30, a few rows of tens of thousands of entries (many bufferfuls of a wave)
        lens = (10.0 * rng.random(rows) ** (-1.0 / 1.5)).astype(np.int64)
        lens[rng.integers(0, rows, size=3)] = rng.integers(20000, 70000, size=3)
        return np.minimum(lens, 70000)
    if kind == "mostly-empty":
        lens = np.zeros(rows, dtype=np.int64)
        lens[rng.integers(0, rows, size=max(rows // 20, 1))] = rng.integers(1, 50)
        return lens
    raise ValueError(kind)


KINDS = ["tiny", "stencil-like", "uniform", "heavy-tail", "long", "mostly-empty", "power-law"]


@pytest.mark.parametrize("suf", ["f32", "f64"])
@pytest.mark.parametrize("kind", KINDS)
def test_spmv_random_structures(gpu_ctx, suf, kind):
    npdt, tdt, tol = DT[suf]
    for seed in range(6):
        rng = np.random.default_rng(1000 * KINDS.index(kind) + seed)
        rows = int(rng.integers(1, 40 if kind == "long" else 3000))
        if kind == "power-law":
            rows = int(rng.integers(300, 2000))
        n_cols = int(rng.integers(1, 4000))
        lens = row_lengths(rng, kind, rows)
        pre, post = int(rng.integers(0, 50)), int(rng.integers(0, 50))  # the piece sits inside larger arrays
        k_base = int(rng.integers(0, 1000))
        starts = np.concatenate([[0], np.cumsum(lens)]) + k_base + pre
        nnz = int(lens.sum()) + pre + post
        rects = np.stack([starts[:-1], starts[1:] - 1], axis=1).astype(np.int64)
        col = rng.integers(0, n_cols, size=nnz).astype(np.int64)
        val = rng.standard_normal(nnz).astype(npdt)
        # input window: sometimes the whole column range, sometimes a narrower one (guards)
        if rng.random() < 0.5:
            x_lo, x_hi = 0, n_cols - 1
        else:
            x_lo = int(rng.integers(0, n_cols))
            x_hi = int(rng.integers(x_lo, n_cols))
        x = rng.standard_normal(x_hi - x_lo + 1).astype(npdt)
        y0 = rng.standard_normal(rows).astype(npdt)
        beta = int(rng.integers(0, 2))
        print(f"case fuzz csr {suf} {kind} seed={seed} rows={rows} nnz={nnz} k_base={k_base} "
              f"x=[{x_lo},{x_hi}] beta={beta}", flush=True)
        want = y0.copy()
        oracle.csr_matvec(rects, k_base, col, val, x, x_lo, x_hi, want, beta=beta)
        bound = np.zeros(rows, dtype=npdt)
        oracle.csr_matvec(rects, k_base, col, np.abs(val), np.abs(x), x_lo, x_hi, bound, beta=0)
        lim = tol * (bound.astype(np.float64) + np.abs(y0) + 1e-300)
        rd, cd, vd, xd = dev(rects), dev(col), dev(val), dev(x)
        for variant in (1, 0):
            _lib.call("lsk_set_option", b"csr.variant", variant)
            yd = dev(y0)
            _lib.call(f"lsk_csr_spmv_{suf}", rows, nnz, P(rd), k_base, P(cd), P(vd), P(xd), x_lo, x_hi, P(yd),
                      beta, stream())
            err = np.abs(yd.cpu().numpy().astype(np.float64) - want.astype(np.float64))
            assert np.all(err <= lim), (variant, float(err.max()), int(np.argmax(err - lim)))
        _lib.call("lsk_set_option", b"csr.variant", 1)
        # fused product + dot on the same piece
        w = rng.standard_normal(rows).astype(npdt)
        slots = _lib.lib().lsk_csr_spmv_dot_partials(rows, nnz, np.dtype(npdt).itemsize)
        partials = torch.zeros(slots, dtype=tdt, device="cuda")
        count = ctypes.c_int64(0)
        yd, wd = dev(y0), dev(w)
        _lib.call(f"lsk_csr_spmv_dot_{suf}", rows, nnz, P(rd), k_base, P(cd), P(vd), P(xd), x_lo, x_hi, P(yd),
                  beta, P(wd), P(partials), slots, ctypes.byref(count), stream())
        out = torch.zeros(1, dtype=tdt, device="cuda")
        _lib.call(f"lsk_sum_partials_{suf}", count.value, P(partials), P(out), None, 0, stream())
        yh = yd.cpu().numpy().astype(np.float64)
        assert np.all(np.abs(yh - want.astype(np.float64)) <= lim)
        assert abs(float(out) - float(np.sum(yh * w))) <= (1e-12 if suf == "f64" else 1e-5) * float(np.sum(np.abs(yh * w))) + 1e-300
        # the same piece on the set-up-time 32-bit structure: bit-identical y and partial sums
        got64, part64 = yd.clone(), partials.clone()
        c32 = torch.empty(nnz, dtype=torch.int32, device="cuda")
        off32 = torch.empty(rows + 1, dtype=torch.int32, device="cuda")
        _lib.call("lsk_csr_compress_cols_u32", nnz, P(cd), x_lo, x_hi, P(c32), stream())
        _lib.call("lsk_csr_compress_rows_u32", rows, nnz, P(rd), k_base, P(off32), stream())
        yd = dev(y0)
        _lib.call(f"lsk_csr_spmv_c32_{suf}", rows, nnz, P(off32), P(c32), P(vd), P(xd), x_hi - x_lo + 1, P(yd),
                  beta, stream())
        assert torch.equal(yd, got64), "32-bit structure changes the product"
        yd = dev(y0)
        partials.zero_()
        _lib.call(f"lsk_csr_spmv_c32_dot_{suf}", rows, nnz, P(off32), P(c32), P(vd), P(xd), x_hi - x_lo + 1, P(yd),
                  beta, P(wd), P(partials), slots, ctypes.byref(count), stream())
        assert torch.equal(yd, got64) and torch.equal(partials, part64)
        # rows too long for one wave: the split product (one workgroup per chunk of such a row, chunk sums
        # added in chunk order) -- every other row bit for bit as above, the long ones to the oracle's bar,
        # the same bits on every call, and a run of rows from the middle of the piece
        plan = ctypes.c_void_p(0)
        code = _lib.lib().lsk_csr_long_rows_create(rows, P(off32), 0, stream(), ctypes.byref(plan))
        assert code == (_lib.LSK_OK if lens.max() > 8192 else _lib.LSK_ERR_UNSUPPORTED), (code, int(lens.max()))
        if code == _lib.LSK_OK:
            try:
                info = (ctypes.c_int64 * 4)()
                _lib.call("lsk_csr_long_rows_info", plan, info)
                long_mask = lens > 8192
                assert (info[0], info[1]) == (int(long_mask.sum()), int(((lens[long_mask] + 8191) // 8192).sum()))
                split_y = []
                for _ in range(2):
                    yd = dev(y0)
                    _lib.call(f"lsk_csr_spmv_c32_split_{suf}", plan, 0, rows, nnz, P(off32), P(c32), P(vd), P(xd),
                              x_hi - x_lo + 1, P(yd), beta, stream())
                    split_y.append(yd)
                assert torch.equal(split_y[0], split_y[1])
                short = torch.from_numpy(~long_mask).cuda()
                assert torch.equal(split_y[0][short], got64[short])
                err = np.abs(split_y[0].cpu().numpy().astype(np.float64) - want.astype(np.float64))
                assert np.all(err <= lim), ("split", float(err.max()), int(np.argmax(err - lim)))
                ra, rb = rows // 3, rows - rows // 4
                yd = dev(y0)
                _lib.call(f"lsk_csr_spmv_c32_split_{suf}", plan, ra, rb - ra, int(lens[ra:rb].sum()), P(off32[ra:]), P(c32),
                          P(vd), P(xd), x_hi - x_lo + 1, P(yd[ra:]), beta, stream())
                # (a run of rows has tiles of its own, so its ordinary rows agree with the whole product's to
                # rounding; the listed rows are cut into the same chunks either way)
                in_run = torch.from_numpy(long_mask & (np.arange(rows) >= ra) & (np.arange(rows) < rb)).cuda()
                assert torch.equal(yd[in_run], split_y[0][in_run])
                err = np.abs(yd[ra:rb].cpu().numpy().astype(np.float64) - want[ra:rb].astype(np.float64))
                assert np.all(err <= lim[ra:rb]), ("split run", float(err.max()))
                assert torch.equal(yd[:ra].cpu(), torch.from_numpy(y0[:ra])) and torch.equal(yd[rb:].cpu(), torch.from_numpy(y0[rb:]))
            finally:
                _lib.call("lsk_csr_long_rows_destroy", plan)
        # the same entries as COO (row-sorted, then shuffled), accumulating into y0
        live = lens.sum()
        if live:
            r_coo = np.repeat(np.arange(rows, dtype=np.int64), lens)
            sl = slice(pre, pre + live)
            for order in (np.arange(live), rng.permutation(live)):
                rc, cc, vc = r_coo[order].copy(), col[sl][order].copy(), val[sl][order].copy()
                want_c = y0.copy()
                oracle.coo_matvec(rc, cc, vc, x, x_lo, x_hi, want_c, 0, rows - 1)
                rcd, ccd, vcd = dev(rc), dev(cc), dev(vc)  # kept alive across the asynchronous launch
                for variant in (1, 0):
                    _lib.call("lsk_set_option", b"coo.variant", variant)
                    yd = dev(y0)
                    _lib.call(f"lsk_coo_spmv_{suf}", int(live), P(rcd), P(ccd), P(vcd), P(xd), x_lo, x_hi,
                              P(yd), 0, rows - 1, stream())
                    torch.cuda.synchronize()
                    err = np.abs(yd.cpu().numpy().astype(np.float64) - want_c.astype(np.float64))
                    assert np.all(err <= lim * 4), ("coo", variant, float(err.max()))
                _lib.call("lsk_set_option", b"coo.variant", 1)
                # 32-bit index fields, rows counted from a random origin at or below row 0's id
                r32 = torch.empty(int(live), dtype=torch.int32, device="cuda")
                c32c = torch.empty(int(live), dtype=torch.int32, device="cuda")
                _lib.call("lsk_csr_compress_cols_u32", int(live), P(rcd), 0, rows - 1, P(r32), stream())
                _lib.call("lsk_csr_compress_cols_u32", int(live), P(ccd), x_lo, x_hi, P(c32c), stream())
                yd = dev(y0)
                _lib.call(f"lsk_coo_spmv_c32_{suf}", int(live), P(r32), P(c32c), P(vcd), P(xd), x_hi - x_lo + 1,
                          P(yd), 0, rows - 1, stream())
                torch.cuda.synchronize()
                err = np.abs(yd.cpu().numpy().astype(np.float64) - want_c.astype(np.float64))
                assert np.all(err <= lim * 4), ("coo 32-bit", float(err.max()))
            _lib.call("lsk_set_option", b"coo.variant", 1)


@pytest.mark.parametrize("values", ["random", "by-distance"])
@pytest.mark.parametrize("seed", range(int(__import__("os").environ.get("LSK_FUZZ_SEEDS", "8"))))
@pytest.mark.parametrize("suf", ["f64", "f32"])
def test_sliced_pattern_products_fuzz(seed, suf, values):
    """Random matrices made of a few row patterns (lists of column distances with repeats, in any
    order, some of them ordered parts of others, some unrelated, some empty), laid out in runs of
    random length so that blocks come out uniform, masked and mixed: the sliced-pattern product equals
    the oracle's ascending-k loop bit for bit, whole and by ranges of rows.  Odd seeds draw lists made
    of a few runs of consecutive distances (served through the LDS staging area), even seeds arbitrary
    ones (mostly too many runs for it: x straight from memory).  values = "by-distance": an entry's value
    depends on its distance alone, so that every position of every list has one value, and option
    sliced.uniform is on: where the structure qualifies the product takes the values as scalars."""
    npdt, tdt, _ = DT[suf]
    rng = np.random.default_rng(1000 + seed)
    rows = int(rng.integers(65, 6000))
    n_patterns = int(rng.integers(1, 9))
    if seed % 2:
        nr = int(rng.integers(1, 6))
        base = np.concatenate([int(s0) + np.arange(int(ln)) for s0, ln in
                               zip(rng.integers(-40, 36, size=nr), rng.integers(1, 7, size=nr))])[:32]
    else:
        base = rng.integers(-40, 41, size=int(rng.integers(8, 33)))      # one long list ...
    patterns = [base]
    for _ in range(n_patterns - 1):
        kind = rng.integers(0, 8)
        if kind < 4:       # ... ordered parts of it, a few entries short (masked slices)
            keep = np.sort(rng.choice(base.size, size=max(base.size - int(rng.integers(1, 4)), 0), replace=False))
            patterns.append(base[keep])
        elif kind < 7:     # ... unrelated lists of about the same length (mixed slices)
            patterns.append(rng.integers(-40, 41, size=max(base.size - int(rng.integers(0, 3)), 1)))
        else:              # ... the empty row
            patterns.append(np.zeros(0, dtype=np.int64))
    run = int(rng.choice([1, 7, 64, 200]))
    which = np.repeat(rng.integers(0, len(patterns), size=rows // run + 1), run)[:rows]
    lens = np.array([patterns[p].size for p in which])
    row_first = int(rng.integers(0, 500))
    k_base = int(rng.integers(0, 100))
    hi = k_base + np.cumsum(lens) - 1
    rects = np.stack([hi - lens + 1, hi], axis=1).astype(np.int64)
    col = np.concatenate([row_first + r + patterns[p] for r, p in enumerate(which)] + [np.zeros(0, dtype=np.int64)])
    col = col.astype(np.int64)
    nnz = col.size
    if nnz == 0:
        pytest.skip("drew an empty matrix")
    val = rng.standard_normal(nnz).astype(npdt)
    if values == "by-distance":
        of_distance = rng.standard_normal(81).astype(npdt)
        val = of_distance[col - (row_first + np.repeat(np.arange(rows), lens)) + 40]
    x_lo, x_hi = int(col.min()), int(col.max())
    x = rng.standard_normal(x_hi - x_lo + 1).astype(npdt)
    print(f"case sliced fuzz seed={seed} {suf} rows={rows} nnz={nnz} patterns={len(patterns)} run={run}", flush=True)
    rd, cd, vd, xd = dev(rects), dev(col), dev(val), dev(x)
    handle = ctypes.c_void_p(0)
    _lib.call("lsk_set_option", b"sliced.uniform", 1 if values == "by-distance" else 0)
    code = getattr(_lib.lib(), f"lsk_csr_sliced_create_{suf}")(rows, nnz, P(rd), k_base, P(cd), P(vd), row_first, x_lo,
                                                             x_hi, stream(), ctypes.byref(handle))
    if code != _lib.LSK_OK:
        _lib.call("lsk_set_option", b"sliced.uniform", 0)
    if code == _lib.LSK_ERR_UNSUPPORTED:   # too much padding for this draw: a legitimate answer
        br = 512 if suf == "f64" else 1024
        padded = sum(int(lens[s:s + br].size) * int(lens[s:s + br].max()) for s in range(0, rows, br))
        assert padded * np.dtype(npdt).itemsize > nnz * (np.dtype(npdt).itemsize + 3)
        return
    assert code == _lib.LSK_OK, _lib.lib().lsk_last_error().decode()
    try:
        info = (ctypes.c_int64 * 9)()
        _lib.call("lsk_csr_sliced_info", handle, info)
        assert info[3] <= len(patterns) and info[2] <= info[1] and info[4] <= 4 * (info[1] - info[2])
        print(f"  blocks {info[1]}: {info[2]} mixed, {info[4]} masked waves; {info[3]} patterns, "
              f"{info[6]} bytes of LDS; uniform values {_lib.lib().lsk_csr_sliced_uniform_values(handle)}", flush=True)
        y0 = rng.standard_normal(rows).astype(npdt)
        for beta, (ra, rb) in ((0, (0, rows)), (1, (0, rows)), (1, (rows // 4, rows - rows // 3)), (0, (63, 130)),
                               (0, (509, 1030)), (1, (1023, 1025))):
            ra = min(ra, rows)
            rb = min(rb, rows)
            want = y0.copy()
            part = want[ra:rb].copy()
            oracle.csr_matvec(rects[ra:rb], k_base, col, val, x, x_lo, x_hi, part, beta=beta)
            want[ra:rb] = part
            yd = dev(y0)
            _lib.call(f"lsk_csr_sliced_spmv_{suf}", handle, ra, rb, P(xd), P(yd), beta, stream())
            got = yd.cpu().numpy()
            assert np.array_equal(got.view(np.uint8), want.view(np.uint8)), \
                f"seed {seed}: beta={beta} rows {ra}..{rb}: {int((got != want).sum())} rows differ (info {list(info)})"
        # the other kernel forms on the whole piece: block-staged, wave-staged through registers, and (where
        # every block runs on one staged list) two right-hand sides in one pass
        want = np.zeros(rows, dtype=npdt)
        oracle.csr_matvec(rects, k_base, col, val, x, x_lo, x_hi, want, beta=0)
        for form, dma in ((0, 1), (1, 0), (1, 1)):
            _lib.call("lsk_set_option", b"sliced.form", form)
            _lib.call("lsk_set_option", b"sliced.dma", dma)
            yd = dev(y0)
            _lib.call(f"lsk_csr_sliced_spmv_{suf}", handle, 0, rows, P(xd), P(yd), 0, stream())
            assert np.array_equal(yd.cpu().numpy().view(np.uint8), want.view(np.uint8)), f"seed {seed}: form {form} dma {dma}"
            if info[7]:
                xb = rng.standard_normal(x.size).astype(npdt)
                want_b = np.zeros(rows, dtype=npdt)
                oracle.csr_matvec(rects, k_base, col, val, xb, x_lo, x_hi, want_b, beta=0)
                ya, yb, xbd = dev(y0), dev(y0), dev(xb)
                _lib.call(f"lsk_csr_sliced_spmv2_{suf}", handle, 0, rows, P(xd), P(xbd), P(ya), P(yb), 0, stream())
                assert np.array_equal(ya.cpu().numpy().view(np.uint8), want.view(np.uint8)) and \
                    np.array_equal(yb.cpu().numpy().view(np.uint8), want_b.view(np.uint8)), f"seed {seed}: pair, form {form} dma {dma}"
    finally:
        _lib.call("lsk_set_option", b"sliced.form", 1)
        _lib.call("lsk_set_option", b"sliced.dma", 1)
        _lib.call("lsk_set_option", b"sliced.uniform", 0)
        _lib.call("lsk_csr_sliced_destroy", handle)
