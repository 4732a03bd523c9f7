"""Scalar: the reference's future-wrapped scalar (src/Scalar.{hpp,cpp}).

A Legion::Future<T> becomes one T in HBM, written by a stream-ordered kernel.  Arithmetic
operators launch the one-thread scalar kernels that replace the reference's scalar tasks
(src/UtilityTasks.cpp:34-99); nothing here synchronises except get_value().
"""
from __future__ import annotations

import torch

from . import _lib


class Scalar:
    def __init__(self, ctx, value, dtype: torch.dtype = None):
        """Scalar(ctx, tensor) wraps a 1-element device tensor (a future);
        Scalar(ctx, number, dtype) makes a ready future (Future::from_value)."""
        self.ctx = ctx
        if isinstance(value, torch.Tensor):
            if value.numel() != 1:
                raise ValueError("a Scalar wraps exactly one element")
            self.future = value
        else:
            if dtype is None:
                raise ValueError("Scalar from a number needs its entry type")
            self.future = ctx.backend.new_scalar(value, dtype)

    @property
    def dtype(self) -> torch.dtype:
        return self.future.dtype

    def get_future(self) -> torch.Tensor:
        return self.future

    def get_value(self):
        """Blocks until the value is ready (Future::get_result)."""
        return self.future.item()

    def _op(self, op: int, other=None) -> "Scalar":
        out = self.ctx.backend.empty(1, self.dtype)
        b = None
        if other is not None:
            if not isinstance(other, Scalar):
                other = Scalar(self.ctx, other, self.dtype)
            b = other.future
        self.ctx.backend.scalar_op(op, self.future, b, out)
        return Scalar(self.ctx, out)

    def __pos__(self):
        return self

    def __neg__(self):
        return self._op(_lib.SOP_NEG)

    def __add__(self, rhs):
        return self._op(_lib.SOP_ADD, rhs)

    def __sub__(self, rhs):
        return self._op(_lib.SOP_SUB, rhs)

    def __mul__(self, rhs):
        return self._op(_lib.SOP_MUL, rhs)

    def __truediv__(self, rhs):
        return self._op(_lib.SOP_DIV, rhs)

    def sqrt(self):
        return self._op(_lib.SOP_SQRT)

    def rsqrt(self):
        return self._op(_lib.SOP_RSQRT)

    def print(self, dummy=None):
        """PrintScalarTask (src/UtilityTasks.cpp:25-31): prints the value, returns a token
        that orders successive prints."""
        v = self.get_value()
        # std::cout << value formats with 6 significant digits ("%g").
        print(f"{v:g}", flush=True)
        return 0
