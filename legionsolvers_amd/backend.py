"""Leaf-operation backend: every vector, matrix and scalar operation of the host mirror ends
in one method here, and each method is one call through the C ABI of include/lsk.h on raw
device pointers and the current torch stream.

`HipBackend` is the only backend the package contains.  It refuses to exist without a GPU
and the built library: there is no CPU fallback in the product path.
"""
from __future__ import annotations

import ctypes
import os
import threading
import weakref
from typing import Optional, Sequence, Tuple

import torch

from . import _lib


def _ptr(t) -> ctypes.c_void_p:
    return ctypes.c_void_p(0 if t is None else t.data_ptr())


class HipBackend:
    name = "hip"

    def __init__(self, device: torch.device):
        if not torch.cuda.is_available():
            raise RuntimeError("HipBackend needs a GPU: torch.cuda.is_available() is False and "
                               "there is no CPU fallback")
        self.lib = _lib.lib()
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError(f"HipBackend needs a cuda (HIP) device, got {self.device}")
        index = self.device.index if self.device.index is not None else torch.cuda.current_device()
        self.device = torch.device("cuda", index)
        torch.cuda.set_device(index)
        _lib.check(self.lib.lsk_init(index))
        # LSK_UNIFORM=1: the library's option "sliced.uniform" (a constant-coefficient stencil's values as
        # scalars; off by default) for programs that cannot set it themselves
        if os.environ.get("LSK_UNIFORM") in ("0", "1"):
            _lib.call("lsk_set_option", b"sliced.uniform", int(os.environ["LSK_UNIFORM"]))
        # Partial sums of dot/nrm2: private to this backend so that other users of the
        # library's per-device scratch cannot race with it.
        self._ws_bytes = int(self.lib.lsk_dot_workspace_bytes(0))
        self._placed = threading.local()  # the library's xcd_remap option (kept per host thread) as last set
        self._ws = torch.empty(self._ws_bytes, dtype=torch.uint8, device=self.device)

    # -- plumbing ------------------------------------------------------------------------
    def stream(self) -> ctypes.c_void_p:
        """torch's current stream on this device, as the raw hipStream_t (asked for on every launch,
        so through the one-call accessor rather than a torch.cuda.Stream object)."""
        return ctypes.c_void_p(torch._C._cuda_getCurrentRawStream(self.device.index))

    def empty(self, n: int, dtype: torch.dtype) -> torch.Tensor:
        """Uninitialised device memory for n elements.  When the device has no room for it, the matrices that
        keep set-up-time copies are asked to give them up, one at a time (register_releasable), and the
        allocation is tried again: the copies only make products faster, a vector that cannot be allocated
        ends the solve."""
        while True:
            try:
                return torch.empty(int(n), dtype=dtype, device=self.device)
            except torch.cuda.OutOfMemoryError:
                if not self.release_optional_memory():
                    raise

    def register_releasable(self, owner) -> None:
        """`owner.release_optional_memory() -> bool` frees some device memory its owner can work without and
        says whether it did; called when an allocation of this backend fails.  Held weakly."""
        if getattr(self, "_releasable", None) is None:
            self._releasable = weakref.WeakSet()
        self._releasable.add(owner)

    def release_optional_memory(self) -> bool:
        if getattr(self, "recordings_alive", 0) > 0:
            return False   # (a recorded step refers to the copies its products read: they stay)
        if getattr(self, "optional_memory_stays", False):
            return False   # (the caller has something else to give up first: SquarePlanner.allocate_workspace)
        for owner in list(getattr(self, "_releasable", None) or ()):
            if owner.release_optional_memory():
                return True
        return False

    def new_scalar(self, value, dtype: torch.dtype) -> torch.Tensor:
        return torch.full((1,), value, dtype=dtype, device=self.device)

    def synchronize(self) -> None:
        torch.cuda.current_stream(self.device).synchronize()

    def _place(self, placement: int) -> None:
        """Sets the tile -> XCD placement for this thread's next launches."""
        if getattr(self._placed, "value", None) != placement:
            _lib.call("lsk_set_option", b"xcd_remap", placement)
            self._placed.value = placement

    # -- BLAS-1 ----------------------------------------------------------------------------
    @staticmethod
    def _futs(futs: Sequence[torch.Tensor]):
        if len(futs) > 4:
            raise ValueError("alpha takes at most 4 futures")
        ptrs = [_ptr(f) for f in futs] + [ctypes.c_void_p(0)] * (4 - len(futs))
        return ctypes.c_int(len(futs)), ptrs

    def fill(self, x: torch.Tensor, value) -> None:
        _lib.call(f"lsk_fill_{_lib.suffix(x.dtype)}", x.numel(), value, _ptr(x), self.stream())

    def random_fill(self, x: torch.Tensor, seed: int, first_index: int) -> None:
        _lib.call(f"lsk_random_fill_{_lib.suffix(x.dtype)}", x.numel(), seed & 0xFFFFFFFFFFFFFFFF, first_index,
                  _ptr(x), self.stream())

    def copy(self, dst: torch.Tensor, src: torch.Tensor) -> None:
        _lib.call(f"lsk_copy_{_lib.suffix(dst.dtype)}", dst.numel(), _ptr(src), _ptr(dst), self.stream())

    def scal(self, x: torch.Tensor, futs) -> None:
        nf, p = self._futs(futs)
        _lib.call(f"lsk_scal_dev_{_lib.suffix(x.dtype)}", x.numel(), nf, *p, _ptr(x), self.stream())

    def axpy(self, y: torch.Tensor, futs, x: torch.Tensor) -> None:
        nf, p = self._futs(futs)
        _lib.call(f"lsk_axpy_dev_{_lib.suffix(y.dtype)}", y.numel(), nf, *p, _ptr(x), _ptr(y), self.stream())

    def xpay(self, y: torch.Tensor, futs, x: torch.Tensor) -> None:
        nf, p = self._futs(futs)
        _lib.call(f"lsk_xpay_dev_{_lib.suffix(y.dtype)}", y.numel(), nf, *p, _ptr(x), _ptr(y), self.stream())

    # -- fused forms -----------------------------------------------------------------------
    SQNORM_SLOTS = 65536  # partial sums lsk_axpy_sqnorm_* / lsk_axpy_dot_dev_* write at most (LSK_MAX_PARTIALS)

    def sqnorm_slots(self, t: torch.Tensor) -> int:
        """Room to offer for the partial sums of a fused update of `t`: 2048 always suffice; a vector that no
        cache holds is updated a few per cent faster with one workgroup (one partial sum) per 16 KiB."""
        return min(max(t.numel() * t.element_size() // 16384 + 1, 2048), self.SQNORM_SLOTS)

    def axpy_sqnorm(self, y: torch.Tensor, futs, x: torch.Tensor, partials: torch.Tensor) -> int:
        """y <- fma(alpha, x, y); partials[:count] <- shares of sum(y_new^2).  Returns count."""
        nf, p = self._futs(futs)
        count = ctypes.c_int(0)
        _lib.call(f"lsk_axpy_sqnorm_dev_{_lib.suffix(y.dtype)}", y.numel(), nf, *p, _ptr(x), _ptr(y),
                  _ptr(partials), partials.numel(), ctypes.byref(count), self.stream())
        return count.value

    def axpy_dot(self, y: torch.Tensor, futs, x: torch.Tensor, w, partials: torch.Tensor) -> int:
        """y <- fma(alpha, x, y); partials[:count] <- shares of sum(y_new * w) (w None: of sum(y_new^2)).
        Returns count."""
        if w is None:
            return self.axpy_sqnorm(y, futs, x, partials)
        nf, p = self._futs(futs)
        count = ctypes.c_int(0)
        _lib.call(f"lsk_axpy_dot_dev_{_lib.suffix(y.dtype)}", y.numel(), nf, *p, _ptr(x), _ptr(y), _ptr(w),
                  _ptr(partials), partials.numel(), ctypes.byref(count), self.stream())
        return count.value

    def axpy_xpay(self, z: torch.Tensor, futs_a, p: torch.Tensor, futs_b, r: torch.Tensor) -> None:
        """z <- fma(alpha, p, z) and p <- fma(beta, p, r) from the old p, one pass."""
        nfa, pa = self._futs(futs_a)
        nfb, pb = self._futs(futs_b)
        _lib.call(f"lsk_axpy_xpay_dev_{_lib.suffix(p.dtype)}", p.numel(), nfa, *pa, nfb, *pb, _ptr(r), _ptr(p),
                  _ptr(z), self.stream())

    def csr_spmv_dot_slots(self, rows: int, nnz: int, dtype: torch.dtype) -> int:
        return int(self.lib.lsk_csr_spmv_dot_partials(rows, nnz, 4 if dtype == torch.float32 else 8))

    def csr_spmv_dot(self, rows: int, nnz: int, rects: torch.Tensor, k_base: int, col: torch.Tensor,
                     val: torch.Tensor, x: torch.Tensor, x_lo: int, x_hi: int, y: torch.Tensor,
                     beta: int, w: torch.Tensor, partials: torch.Tensor, placement: int = 0) -> int:
        """y as csr_spmv; partials[:count] <- shares of sum(y_new * w).  Returns count."""
        self._place(placement)
        count = ctypes.c_int64(0)
        if col.dtype == torch.int32:
            _lib.call(f"lsk_csr_spmv_c32_dot_{_lib.suffix(val.dtype)}", rows, nnz, _ptr(rects), _ptr(col),
                      _ptr(val), _ptr(x), x_hi - x_lo + 1, _ptr(y), beta, _ptr(w), _ptr(partials),
                      partials.numel(), ctypes.byref(count), self.stream())
            return count.value
        _lib.call(f"lsk_csr_spmv_dot_{_lib.suffix(val.dtype)}", rows, nnz, _ptr(rects), k_base, _ptr(col),
                  _ptr(val), _ptr(x), x_lo, x_hi, _ptr(y), beta, _ptr(w), _ptr(partials), partials.numel(),
                  ctypes.byref(count), self.stream())
        return count.value

    MULTI_MAX = 64  # vectors per lsk_multi_dot_* / lsk_multi_axpy_* call

    def multi_dot(self, vs: Sequence[torch.Tensor], w: torch.Tensor, out: torch.Tensor) -> None:
        """out[j] <- vs[j] . w for every j, in one pass over w per group of 8 vectors."""
        k = len(vs)
        need = int(self.lib.lsk_multi_dot_workspace_bytes(k))
        if getattr(self, "_multi_ws", None) is None or self._multi_ws.numel() < need:
            self._multi_ws = torch.empty(need, dtype=torch.uint8, device=self.device)
        ptrs = (ctypes.c_void_p * max(k, 1))(*[v.data_ptr() for v in vs])
        _lib.call(f"lsk_multi_dot_{_lib.suffix(w.dtype)}", w.numel(), k, ptrs, _ptr(w), _ptr(out),
                  _ptr(self._multi_ws), self._multi_ws.numel(), self.stream())

    def multi_axpy(self, y: torch.Tensor, coef: torch.Tensor, xs: Sequence[torch.Tensor], negate: bool,
                   partials=None) -> int:
        """y <- y + sum_j (+-coef[j]) xs[j], as the chain of axpys in ascending j (bit-identical to it);
        with `partials`, also the shares of sum(y_new^2).  Returns their count (0 without)."""
        k = len(xs)
        ptrs = (ctypes.c_void_p * max(k, 1))(*[x.data_ptr() for x in xs])
        count = ctypes.c_int(0)
        _lib.call(f"lsk_multi_axpy_{_lib.suffix(y.dtype)}", y.numel(), k, ptrs, _ptr(coef), 1 if negate else 0, _ptr(y),
                  _ptr(partials), partials.numel() if partials is not None else 0,
                  ctypes.byref(count) if partials is not None else None, self.stream())
        return count.value

    def sum_partials(self, count: int, partials, out: torch.Tensor) -> None:
        """out[0] <- partials[0] + ... + partials[count-1] (count may be 0)."""
        _lib.call(f"lsk_sum_partials_{_lib.suffix(out.dtype)}", count, _ptr(partials), _ptr(out),
                  _ptr(self._ws), self._ws_bytes, self.stream())

    def dot(self, v: torch.Tensor, w: torch.Tensor, out: torch.Tensor) -> None:
        _lib.call(f"lsk_dot_{_lib.suffix(v.dtype)}", v.numel(), _ptr(v), _ptr(w), _ptr(out),
                  _ptr(self._ws), self._ws_bytes, self.stream())

    def sum_scalars(self, inp: torch.Tensor, out: torch.Tensor) -> None:
        _lib.call(f"lsk_sum_scalars_{_lib.suffix(inp.dtype)}", inp.numel(), _ptr(inp), _ptr(out), self.stream())

    def scalar_op(self, op: int, a: torch.Tensor, b, out: torch.Tensor) -> None:
        _lib.call(f"lsk_scalar_op_{_lib.suffix(a.dtype)}", op, _ptr(a), _ptr(b), _ptr(out), self.stream())

    # -- SpMV ------------------------------------------------------------------------------
    def csr_spmv(self, rows: int, nnz: int, rects: torch.Tensor, k_base: int, col: torch.Tensor,
                 val: torch.Tensor, x: torch.Tensor, x_lo: int, x_hi: int, y: torch.Tensor,
                 beta: int, placement: int = 0) -> None:
        """`placement`: the piece's tile -> XCD placement found by csr_spmv_tune (a speed hint).
        `rects`, `col`: the (rows, 2) int64 rects and int64 global column ids, or the int32-typed
        arrays of csr_compress_rows (rows + 1 offsets) and csr_compress_cols (made for the window
        x_lo..x_hi; x[0] is the window's first element either way; k_base is then unused)."""
        self._place(placement)
        if col.dtype == torch.int32:
            _lib.call(f"lsk_csr_spmv_c32_{_lib.suffix(val.dtype)}", rows, nnz, _ptr(rects), _ptr(col),
                      _ptr(val), _ptr(x), x_hi - x_lo + 1, _ptr(y), beta, self.stream())
            return
        _lib.call(f"lsk_csr_spmv_{_lib.suffix(val.dtype)}", rows, nnz, _ptr(rects), k_base, _ptr(col),
                  _ptr(val), _ptr(x), x_lo, x_hi, _ptr(y), beta, self.stream())

    def csr_long_rows_create(self, rowoff: torch.Tensor, threshold: int = 0):
        """The plan for the rows of a piece that are too long for one wave (a LongRows), from the piece's
        rows + 1 32-bit offsets (csr_compress_rows); None when no row is longer than the threshold (0:
        the library's).  Synchronises."""
        handle = ctypes.c_void_p(0)
        code = self.lib.lsk_csr_long_rows_create(rowoff.numel() - 1, _ptr(rowoff), threshold, self.stream(),
                                                 ctypes.byref(handle))
        if code == _lib.LSK_ERR_UNSUPPORTED:
            return None
        _lib.check(code)
        return LongRows(self.lib, handle)

    def csr_spmv_split(self, plan: "LongRows", row_begin: int, rows: int, nnz: int, rowoff: torch.Tensor,
                       col32: torch.Tensor, val: torch.Tensor, x: torch.Tensor, x_lo: int, x_hi: int, y: torch.Tensor,
                       beta: int, placement: int = 0) -> None:
        """csr_spmv on the 32-bit structure for the rows row_begin .. row_begin + rows - 1 of the plan's
        piece (rowoff, y: those rows'), the rows the plan lists multiplied in chunks by workgroups of their own."""
        self._place(placement)
        _lib.call(f"lsk_csr_spmv_c32_split_{_lib.suffix(val.dtype)}", plan.handle, row_begin, rows, nnz, _ptr(rowoff),
                  _ptr(col32), _ptr(val), _ptr(x), x_hi - x_lo + 1, _ptr(y), beta, self.stream())

    def csr_sliced_create(self, rows: int, nnz: int, rects: torch.Tensor, k_base: int, col: torch.Tensor,
                          val: torch.Tensor, row_first: int, x_lo: int, x_hi: int):
        """The sliced-pattern copy of a CSR piece for products against the window x_lo..x_hi (a
        SlicedStructure), or None when the matrix has no such structure.  Synchronises."""
        handle = ctypes.c_void_p(0)
        code = getattr(self.lib, f"lsk_csr_sliced_create_{_lib.suffix(val.dtype)}")(
            rows, nnz, _ptr(rects), k_base, _ptr(col), _ptr(val), row_first, x_lo, x_hi, self.stream(),
            ctypes.byref(handle))
        if code == _lib.LSK_ERR_UNSUPPORTED:
            return None
        _lib.check(code)
        return SlicedStructure(self.lib, handle, val.dtype)

    def csr_sliced_update_values(self, sliced: "SlicedStructure", rows: int, nnz: int, rects: torch.Tensor,
                                 k_base: int, val: torch.Tensor) -> None:
        """Copies the entry values into the sliced-pattern copy again (same structure, new values)."""
        _lib.call(f"lsk_csr_sliced_update_values_{_lib.suffix(sliced.dtype)}", sliced.handle, rows, nnz, _ptr(rects),
                  k_base, _ptr(val), self.stream())

    def csr_sliced_spmv(self, sliced: "SlicedStructure", row_begin: int, row_end: int, x: torch.Tensor,
                        y: torch.Tensor, beta: int, placement: int = 0) -> None:
        """Rows row_begin..row_end-1 of y <- A x + beta y; x[0] is the window's first element, y[0] the
        piece's first row."""
        self._place(placement)
        _lib.call(f"lsk_csr_sliced_spmv_{_lib.suffix(sliced.dtype)}", sliced.handle, row_begin, row_end, _ptr(x),
                  _ptr(y), beta, self.stream())

    def csr_sliced_tune(self, sliced: "SlicedStructure", x: torch.Tensor, y: torch.Tensor) -> Tuple[int, float]:
        """(placement, milliseconds per launch) of the fastest workgroup -> XCD placement; y is
        overwritten.  Synchronises."""
        remap, ms = ctypes.c_int(0), ctypes.c_float(0.0)
        _lib.call(f"lsk_csr_sliced_tune_{_lib.suffix(sliced.dtype)}", sliced.handle, _ptr(x), _ptr(y), self.stream(),
                  ctypes.byref(remap), ctypes.byref(ms))
        return remap.value, ms.value

    def csr_sliced_tune2(self, sliced: "SlicedStructure", xs, ys) -> Optional[Tuple[int, float]]:
        """The same for the two-vector product (ys[k] <- A xs[k], overwritten); None where no two-vector
        form serves the structure.  Synchronises."""
        remap, ms = ctypes.c_int(0), ctypes.c_float(0.0)
        code = getattr(self.lib, f"lsk_csr_sliced_tune2_{_lib.suffix(sliced.dtype)}")(
            sliced.handle, _ptr(xs[0]), _ptr(xs[1]), _ptr(ys[0]), _ptr(ys[1]), self.stream(), ctypes.byref(remap),
            ctypes.byref(ms))
        if code == _lib.LSK_ERR_UNSUPPORTED:
            return None
        _lib.check(code)
        return remap.value, ms.value

    @staticmethod
    def csr_sliced_dot_slots(sliced: "SlicedStructure", row_end: int, row_begin: int = 0) -> int:
        """Partial sums lsk_csr_sliced_spmv_dot_* leaves for rows row_begin..row_end-1: one per block
        of the structure those rows touch."""
        if row_end <= row_begin:
            return 0
        return -(-row_end // sliced.block_rows) - row_begin // sliced.block_rows

    def csr_sliced_spmv_dot(self, sliced: "SlicedStructure", row_end: int, x: torch.Tensor, y: torch.Tensor, beta: int,
                            w: torch.Tensor, partials: torch.Tensor, placement: int = 0, row_begin: int = 0) -> int:
        """Rows row_begin..row_end-1 of the piece's product, and partials[:count] <- shares of the sum
        of y_new * w over those rows.  Returns count."""
        self._place(placement)
        count = ctypes.c_int64(0)
        _lib.call(f"lsk_csr_sliced_spmv_dot_{_lib.suffix(sliced.dtype)}", sliced.handle, row_begin, row_end, _ptr(x),
                  _ptr(y), beta, _ptr(w), _ptr(partials), partials.numel(), ctypes.byref(count), self.stream())
        return count.value

    def csr_sliced_spmv2(self, sliced: "SlicedStructure", rows: int, xs, ys, beta: int, dot_with=None, partials=None,
                         placement: int = 0, row_begin: int = 0):
        """Two products with one pass over the values: ys[k] <- A xs[k] + beta ys[k], k = 0, 1, over the rows
        row_begin..rows-1 of the piece (ys, dot_with: the whole piece's vectors), each bit for bit what
        csr_sliced_spmv gives for it alone.  With dot_with = (w0, w1) and partials = (p0, p1) also the
        shares of ys[k] . w_k over those rows; returns their count (0 without).  None when the structure has
        blocks the two-vector kernel does not serve (the caller then launches the products one by one)."""
        self._place(placement)
        suffix = _lib.suffix(sliced.dtype)
        if dot_with is None:
            code = getattr(self.lib, f"lsk_csr_sliced_spmv2_{suffix}")(
                sliced.handle, row_begin, rows, _ptr(xs[0]), _ptr(xs[1]), _ptr(ys[0]), _ptr(ys[1]), beta, self.stream())
            count = ctypes.c_int64(0)
        else:
            count = ctypes.c_int64(0)
            code = getattr(self.lib, f"lsk_csr_sliced_spmv2_dot_{suffix}")(
                sliced.handle, row_begin, rows, _ptr(xs[0]), _ptr(xs[1]), _ptr(ys[0]), _ptr(ys[1]), beta, _ptr(dot_with[0]),
                _ptr(dot_with[1]), _ptr(partials[0]), _ptr(partials[1]), min(partials[0].numel(), partials[1].numel()),
                ctypes.byref(count), self.stream())
        if code == _lib.LSK_ERR_UNSUPPORTED:
            return None
        _lib.check(code)
        return count.value

    def csr_spmv_tune(self, rows: int, nnz: int, rects: torch.Tensor, k_base: int, col: torch.Tensor,
                      val: torch.Tensor, x: torch.Tensor, x_lo: int, x_hi: int,
                      y: torch.Tensor) -> Tuple[int, float]:
        """(placement, milliseconds per launch) of the fastest tile -> XCD placement for this
        piece; y is overwritten.  Synchronises."""
        remap, ms = ctypes.c_int(0), ctypes.c_float(0.0)
        if col.dtype == torch.int32:
            _lib.call(f"lsk_csr_spmv_c32_tune_{_lib.suffix(val.dtype)}", rows, nnz, _ptr(rects), _ptr(col),
                      _ptr(val), _ptr(x), x_hi - x_lo + 1, _ptr(y), self.stream(), ctypes.byref(remap),
                      ctypes.byref(ms))
            return remap.value, ms.value
        _lib.call(f"lsk_csr_spmv_tune_{_lib.suffix(val.dtype)}", rows, nnz, _ptr(rects), k_base, _ptr(col),
                  _ptr(val), _ptr(x), x_lo, x_hi, _ptr(y), self.stream(), ctypes.byref(remap), ctypes.byref(ms))
        return remap.value, ms.value

    def coo_spmv(self, nnz: int, row: torch.Tensor, col: torch.Tensor, val: torch.Tensor,
                 x: torch.Tensor, x_lo: int, x_hi: int, y: torch.Tensor, y_lo: int, y_hi: int) -> None:
        if row.dtype == torch.int32:  # 32-bit copies: y_lo / y_hi count from the origin of row32
            _lib.call(f"lsk_coo_spmv_c32_{_lib.suffix(val.dtype)}", nnz, _ptr(row), _ptr(col), _ptr(val),
                      _ptr(x), x_hi - x_lo + 1, _ptr(y), y_lo, y_hi, self.stream())
            return
        _lib.call(f"lsk_coo_spmv_{_lib.suffix(val.dtype)}", nnz, _ptr(row), _ptr(col), _ptr(val),
                  _ptr(x), x_lo, x_hi, _ptr(y), y_lo, y_hi, self.stream())

    # -- set-up ------------------------------------------------------------------------------
    COMPRESS_MAX_EXTENT = 0xFFFFFFFF - 1  # widest window 32-bit column offsets can address

    def csr_compress_cols(self, col: torch.Tensor, x_lo: int, x_hi: int) -> torch.Tensor:
        """The column field as 32-bit offsets into the window x_lo..x_hi (bit pattern of
        uint32 in an int32 tensor; 0xffffffff marks a column outside the window)."""
        out = torch.empty(col.numel(), dtype=torch.int32, device=self.device)
        _lib.call("lsk_csr_compress_cols_u32", col.numel(), _ptr(col), x_lo, x_hi, _ptr(out), self.stream())
        return out

    def coo_block_order(self, col32: torch.Tensor, x_extent: int, block_cols: int):
        """Entry order sorted (stably) by column block of `block_cols` columns, as uint32 positions
        in an int32 tensor; None when there are too many entries for it.  Synchronises."""
        order = torch.empty(col32.numel(), dtype=torch.int32, device=self.device)
        code = self.lib.lsk_coo_block_order_u32(col32.numel(), _ptr(col32), x_extent, block_cols, _ptr(order),
                                                self.stream())
        if code == _lib.LSK_ERR_UNSUPPORTED:
            return None
        _lib.check(code)
        return order

    def coo_binned_create(self, row32: torch.Tensor, col32: torch.Tensor, val: torch.Tensor, y_first: int,
                          y_last: int, x_extent: int, block_cols: int = 0, tile_rows: int = 0):
        """The binned copy of a COO piece (a BinnedCoo) for products of the rows y_first..y_last (offsets
        in row32's window) against the first x_extent columns of col32's window, or None when the
        library declines (too many entries or bins, nothing inside the windows).  Synchronises."""
        handle = ctypes.c_void_p(0)
        code = getattr(self.lib, f"lsk_coo_binned_create_{_lib.suffix(val.dtype)}")(
            val.numel(), _ptr(row32), _ptr(col32), _ptr(val), y_first, y_last, x_extent, block_cols, tile_rows,
            self.stream(), ctypes.byref(handle))
        if code == _lib.LSK_ERR_UNSUPPORTED:
            return None
        _lib.check(code)
        return BinnedCoo(self.lib, handle, val.dtype, val.numel())

    def coo_binned_spmv(self, binned: "BinnedCoo", x: torch.Tensor, y: torch.Tensor, beta: int) -> None:
        """y <- A x + beta y through the binned copy; x[0] is the window's first column, y[0] row y_first."""
        _lib.call(f"lsk_coo_binned_spmv_{_lib.suffix(binned.dtype)}", binned.handle, _ptr(x), _ptr(y), beta,
                  self.stream())

    def coo_binned_spmv_phase(self, binned: "BinnedCoo", x: torch.Tensor, y: torch.Tensor, beta: int, phase: int,
                              col_lo: int, col_hi: int) -> None:
        """coo_binned_spmv in two calls: phase 0 multiplies the blocks of columns wholly inside col_lo .. col_hi
        (window-relative; x is read only there), phase 1 the others and writes y."""
        _lib.call(f"lsk_coo_binned_spmv_phase_{_lib.suffix(binned.dtype)}", binned.handle, _ptr(x), _ptr(y), beta, phase,
                  col_lo, col_hi, self.stream())

    def coo_binned_update_values(self, binned: "BinnedCoo", val: torch.Tensor) -> None:
        """Copies the entry values into the binned copy again (same rows and columns, new values)."""
        _lib.call(f"lsk_coo_binned_update_values_{_lib.suffix(binned.dtype)}", binned.handle, val.numel(), _ptr(val),
                  self.stream())

    def gather(self, src: torch.Tensor, order: torch.Tensor) -> torch.Tensor:
        """src[order] for the 32-bit index fields and the entry values."""
        kind = "u32" if src.dtype == torch.int32 else _lib.suffix(src.dtype)
        dst = torch.empty_like(src)
        _lib.call(f"lsk_gather_{kind}", src.numel(), _ptr(src), _ptr(order), _ptr(dst), self.stream())
        return dst

    def csr_entry_rows(self, rects: torch.Tensor, k_base: int, nnz: int) -> torch.Tensor:
        """Row (0-based in `rects`) of every kernel entry, as uint32 in an int32 tensor; 0xffffffff
        for an entry inside no rect."""
        out = torch.empty(nnz, dtype=torch.int32, device=self.device)
        _lib.call("lsk_csr_entry_rows_u32", rects.shape[0], nnz, _ptr(rects), k_base, _ptr(out), self.stream())
        return out

    def sorted_key_offsets(self, keys: torch.Tensor, nkeys: int) -> torch.Tensor:
        """nkeys + 1 offsets: how many of the ascending uint32 `keys` lie below 0, 1, .. nkeys."""
        out = torch.empty(nkeys + 1, dtype=torch.int32, device=self.device)
        _lib.call("lsk_sorted_key_offsets_u32", keys.numel(), _ptr(keys), nkeys, _ptr(out), self.stream())
        return out

    def csr_compress_rows(self, rects: torch.Tensor, k_base: int, nnz: int):
        """rows + 1 32-bit offsets into the piece's col/val arrays (the classic row pointer; bit
        pattern of uint32 in an int32 tensor), or None when the rects are not back to back inside
        the piece or nnz does not fit.  Synchronises."""
        rows = rects.shape[0]
        out = torch.empty(rows + 1, dtype=torch.int32, device=self.device)
        code = getattr(self.lib, "lsk_csr_compress_rows_u32")(rows, nnz, _ptr(rects), k_base, _ptr(out),
                                                             self.stream())
        if code == _lib.LSK_ERR_UNSUPPORTED:
            return None
        _lib.check(code)
        return out

    def csr_halo_rows(self, rows: int, nnz: int, rects: torch.Tensor, k_base: int, col: torch.Tensor,
                      lo: int, hi: int) -> Tuple[int, int]:
        """(largest row with a column below lo or -1, smallest row with a column above hi or
        rows) of a CSR piece.  Synchronises."""
        out = torch.empty(2, dtype=torch.int64, device=self.device)
        _lib.call("lsk_csr_halo_rows_i64", rows, nnz, _ptr(rects), k_base, _ptr(col), lo, hi, _ptr(out),
                  self.stream())
        below, above = out.tolist()
        return int(below), int(above)

    def index_bounds(self, idx: torch.Tensor) -> Tuple[int, int]:
        """(min, max) of an index field; (0, -1) when it is empty.  Synchronises."""
        if idx.numel() == 0:
            return (0, -1)
        out = torch.empty(2, dtype=torch.int64, device=self.device)
        _lib.call("lsk_index_bounds_i64", idx.numel(), _ptr(idx), _ptr(out), self.stream())
        lo, hi = out.tolist()
        return int(lo), int(hi)

    def rect_bounds(self, rects: torch.Tensor) -> Tuple[int, int]:
        """Bounding kernel interval of a block of row rects (rows, 2): the image-range of
        create_partition_by_image_range (src/CSRMatrix.cpp:89-109); (0, -1) if all rows are
        empty.  Set-up time only; synchronises."""
        if rects.numel() == 0:
            return (0, -1)
        lo, hi = rects[:, 0], rects[:, 1]
        live = hi >= lo
        if not bool(live.any()):
            return (0, -1)
        return int(lo[live].min().item()), int(hi[live].max().item())


class LongRows:
    """Owner of one lsk_csr_long_rows_t (device memory held by the library, freed with this object)."""

    def __init__(self, lib, handle):
        self.lib, self.handle = lib, handle
        out = (ctypes.c_int64 * 4)()
        _lib.check(lib.lsk_csr_long_rows_info(handle, out))
        self.long_rows, self.chunks, self.threshold, self.chunk_entries = (int(v) for v in out)

    def close(self) -> None:
        if self.handle:
            self.lib.lsk_csr_long_rows_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class BinnedCoo:
    """Owner of one lsk_coo_binned_t (device memory held by the library, freed with this object)."""

    def __init__(self, lib, handle, dtype, nnz):
        self.lib, self.handle, self.dtype, self.nnz = lib, handle, dtype, nnz
        out = (ctypes.c_int64 * 8)()
        _lib.check(lib.lsk_coo_binned_info(handle, out))
        (self.live, self.padded_entries, self.block_cols, self.tile_rows, self.blocks, self.tiles, self.units,
         self.segments) = (int(v) for v in out)

    def own_bytes(self, rows: int, x_extent: int, elem_size: int) -> int:
        """Bytes one product moves on this layout: pass 1 reads value + column offset and writes the
        product, pass 2 reads product + row offset; x and y once each, 8 per segment, 12 per unit."""
        return (self.padded_entries * (3 * elem_size + 4) + (rows + x_extent) * elem_size + self.segments * 8
                + self.units * 12)

    def close(self) -> None:
        if self.handle:
            self.lib.lsk_coo_binned_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class SlicedStructure:
    """Owner of one lsk_csr_sliced_t (device memory held by the library, freed with this object)."""

    def __init__(self, lib, handle, dtype):
        self.lib, self.handle, self.dtype = lib, handle, dtype
        out = (ctypes.c_int64 * 9)()
        _lib.check(lib.lsk_csr_sliced_info(handle, out))
        (self.padded_entries, self.blocks, self.mixed_blocks, self.patterns, self.masked_waves, self.block_rows,
         self.lds_bytes, pair, runs3) = (int(v) for v in out)
        self.runs_of_three = bool(runs3)  # every list that serves a block is runs of three consecutive distances
        self.pair_served = bool(pair)   # lsk_csr_sliced_spmv2_* serves this structure

    @property
    def uniform_values(self) -> bool:
        """Every position of every list has one value in all rows (asked anew: update_values can change it);
        with option sliced.uniform = 1 the products of such a structure do not read the value array."""
        return bool(self.lib.lsk_csr_sliced_uniform_values(self.handle))

    def algorithmic_bytes(self, rows: int, elem_size: int, without_values: bool = False) -> int:
        """Bytes one product of `rows` rows must move: the padded values (unless the product takes them
        as scalars, see uniform_values), x and y once, 16 per block, a 4-byte mask per row of a masked
        wave, a pattern id per row of a block without a common list."""
        return ((0 if without_values else self.padded_entries * elem_size) + rows * 2 * elem_size + self.blocks * 16
                + self.masked_waves * (self.block_rows // 4) * 4 + self.mixed_blocks * self.block_rows)

    def close(self) -> None:
        if self.handle:
            self.lib.lsk_csr_sliced_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
