"""A leaf backend for CPU-only tests of the HOST logic (vectors, planner, halo exchange,
all-reduce, solver sequences): same method surface as legionsolvers_amd.backend.HipBackend, but
every leaf operation is the CPU oracle on CPU torch tensors.

It lives under tests/ and is injected by tests only; the package itself has no such path.
"""
from __future__ import annotations

import numpy as np
import torch

import oracle
from oracle import partition as part

_OPS = {0: "neg", 1: "add", 2: "sub", 3: "mul", 4: "div", 5: "sqrt", 6: "rsqrt", 7: "copy"}


def _np(t: torch.Tensor) -> np.ndarray:
    return t.numpy()  # shares memory with the (contiguous, CPU) tensor view


class OracleBackend:
    name = "oracle"

    def __init__(self):
        self.device = torch.device("cpu")

    def stream(self):
        return None

    def empty(self, n, dtype):
        return torch.zeros(int(n), dtype=dtype)

    def new_scalar(self, value, dtype):
        return torch.full((1,), value, dtype=dtype)

    def synchronize(self):
        pass

    @staticmethod
    def _alpha(futs, dtype):
        return oracle.get_alpha([f.item() for f in futs], _np(torch.zeros(0, dtype=dtype)).dtype)

    def fill(self, x, value):
        x.fill_(value)

    def random_fill(self, x, seed, first_index):
        x.copy_(torch.from_numpy(oracle.random_fill(x.numel(), seed, first_index, _np(x).dtype)))

    def copy(self, dst, src):
        dst.copy_(src)

    def scal(self, x, futs):
        oracle.scal(self._alpha(futs, x.dtype), _np(x))

    def axpy(self, y, futs, x):
        oracle.axpy(self._alpha(futs, y.dtype), _np(x), _np(y))

    def xpay(self, y, futs, x):
        oracle.xpay(self._alpha(futs, y.dtype), _np(x), _np(y))

    # fused forms: one partial sum each, formed by the oracle's own sequential dot
    SQNORM_SLOTS = 1

    def sqnorm_slots(self, t):
        return 1

    def axpy_sqnorm(self, y, futs, x, partials):
        self.axpy(y, futs, x)
        partials[0] = float(oracle.dot(_np(y), _np(y)))
        return 1

    def axpy_dot(self, y, futs, x, w, partials):
        self.axpy(y, futs, x)
        partials[0] = float(oracle.dot(_np(y), _np(y if w is None else w)))
        return 1

    def axpy_xpay(self, z, futs_a, p, futs_b, r):
        self.axpy(z, futs_a, p)   # z from the old p ...
        self.xpay(p, futs_b, r)   # ... then p

    MULTI_MAX = 64

    def multi_dot(self, vs, w, out):
        for j, v in enumerate(vs):
            out[j] = float(oracle.dot(_np(v), _np(w))) if w.numel() else 0.0

    def multi_axpy(self, y, coef, xs, negate, partials=None):
        c = _np(coef)
        for j, x in enumerate(xs):
            oracle.axpy(-c[j] if negate else c[j], _np(x), _np(y))
        if partials is None:
            return 0
        partials[0] = float(oracle.dot(_np(y), _np(y)))
        return 1

    def csr_spmv_dot_slots(self, rows, nnz, dtype):
        return 1

    def csr_spmv_dot(self, rows, nnz, rects, k_base, col, val, x, x_lo, x_hi, y, beta, w, partials, placement=0):
        self.csr_spmv(rows, nnz, rects, k_base, col, val, x, x_lo, x_hi, y, beta)
        partials[0] = float(oracle.dot(_np(w), _np(y)))
        return 1

    def sum_partials(self, count, partials, out):
        a = _np(partials)[:count] if count else np.zeros(0)
        acc = a.dtype.type(0) if count else 0.0
        for i in range(count):
            acc = acc + a[i]
        out[0] = float(acc)

    def dot(self, v, w, out):
        out[0] = float(oracle.dot(_np(v), _np(w))) if v.numel() else 0.0

    def sum_scalars(self, inp, out):
        a = _np(inp)
        acc = a[0]
        for i in range(1, a.size):
            acc = oracle.scalar_op("add", acc, a[i], a.dtype)
        out[0] = float(acc)

    def scalar_op(self, op, a, b, out):
        bv = b.item() if b is not None else 0.0
        out[0] = float(oracle.scalar_op(_OPS[op], a.item(), bv, _np(a).dtype))

    COMPRESS_MAX_EXTENT = 0xFFFFFFFF - 1

    def csr_compress_cols(self, col, x_lo, x_hi):
        c = _np(col)
        inside = (c >= x_lo) & (c <= x_hi)
        return torch.from_numpy(np.where(inside, c - x_lo, 0xFFFFFFFF).astype(np.uint32).view(np.int32))

    def csr_compress_rows(self, rects, k_base, nnz):
        r = _np(rects)
        if nnz > 0xFFFFFFFF:
            return None
        lo, end = r[:, 0] - k_base, r[:, 1] + 1 - k_base
        if (lo < 0).any() or (end < lo).any() or (end > nnz).any() or (r[1:, 0] != r[:-1, 1] + 1).any():
            return None
        off = np.concatenate([lo, end[-1:]]) if len(r) else np.zeros(1, dtype=np.int64)
        return torch.from_numpy(off.astype(np.uint32).view(np.int32))

    def csr_spmv(self, rows, nnz, rects, k_base, col, val, x, x_lo, x_hi, y, beta, placement=0):
        if rects.dtype == torch.int32:  # rows + 1 offsets from the start of col / val
            off = _np(rects).view(np.uint32).astype(np.int64)
            rects = torch.from_numpy(np.stack([off[:-1], off[1:] - 1], axis=1))
            k_base = 0
        c = _np(col)
        if c.dtype == np.int32:  # compressed: offsets into the window, 0xffffffff = outside it
            u = c.view(np.uint32).astype(np.int64)
            c = np.where(u == 0xFFFFFFFF, x_lo - 1, u + x_lo)
        oracle.csr_matvec(_np(rects), k_base, c, _np(val), _np(x), x_lo, x_hi, _np(y), beta=beta)

    def coo_spmv(self, nnz, row, col, val, x, x_lo, x_hi, y, y_lo, y_hi):
        r, c = _np(row), _np(col)
        if r.dtype == np.int32:  # 32-bit copies: offsets into the row / x windows, 0xffffffff = outside
            ru, cu = r.view(np.uint32).astype(np.int64), c.view(np.uint32).astype(np.int64)
            r = np.where(ru == 0xFFFFFFFF, -1, ru)  # y_lo / y_hi count from the same origin
            c = np.where(cu == 0xFFFFFFFF, x_lo - 1, cu + x_lo)
        oracle.coo_matvec(r, c, _np(val), _np(x), x_lo, x_hi, _np(y), y_lo, y_hi)

    def csr_entry_rows(self, rects, k_base, nnz):
        rows = oracle.csr_entry_rows(_np(rects), k_base, nnz)
        return torch.from_numpy(np.where(rows < 0, 0xFFFFFFFF, rows).astype(np.uint32).view(np.int32))

    def coo_block_order(self, col32, x_extent, block_cols):
        u = _np(col32).view(np.uint32).astype(np.int64)
        last = -(-max(x_extent, 1) // block_cols)
        keys = np.where(u == 0xFFFFFFFF, last, u // block_cols)
        return torch.from_numpy(np.argsort(keys, kind="stable").astype(np.uint32).view(np.int32))

    def gather(self, src, order):
        return src[torch.from_numpy(_np(order).view(np.uint32).astype(np.int64))].contiguous()

    def sorted_key_offsets(self, keys, nkeys):
        u = _np(keys).view(np.uint32).astype(np.int64)
        return torch.from_numpy(np.searchsorted(u, np.arange(nkeys + 1), side="left").astype(np.uint32).view(np.int32))

    def csr_halo_rows(self, rows, nnz, rects, k_base, col, lo, hi):
        r, c = _np(rects), _np(col)
        below, above = -1, rows
        for i in range(rows):
            k0, k1 = max(int(r[i, 0]) - k_base, 0), min(int(r[i, 1]) - k_base, nnz - 1)
            if k1 < k0:
                continue
            seg = c[k0:k1 + 1]
            if (seg < lo).any():
                below = i
            if (seg > hi).any() and above == rows:
                above = i
        return below, above

    def index_bounds(self, idx):
        return part.bounds(_np(idx))

    def rect_bounds(self, rects):
        r = _np(rects)
        live = r[:, 1] >= r[:, 0]
        if not live.any():
            return (0, -1)
        return int(r[live, 0].min()), int(r[live, 1].max())
