"""PartitionedVector: the reference's user-facing vector (src/PartitionedVector.{hpp,cpp}).

A vector is an index space coloured by a disjoint, complete partition; every BLAS-1 method is
one leaf launch per piece (the reference's IndexTaskLauncher over the colour space,
src/PartitionedVector.cpp:195-358).  One process holds the pieces of its rank in ONE
contiguous HBM buffer, in colour order, optionally with room before and after them for the
halo of a 1-D vector, so that an SpMV reads x and its ghost cells through one base pointer
and neighbouring pieces of one rank need no copy at all.
"""
from __future__ import annotations

from typing import List, Tuple

import torch

from .index_space import IndexPartition
from .scalar import Scalar


class PartitionedVector:
    def __init__(self, ctx, name: str, index_partition: IndexPartition,
                 dtype: torch.dtype = torch.float64, pad: Tuple[int, int] = (0, 0)):
        if dtype not in (torch.float32, torch.float64):
            raise TypeError("entry type must be float32 or float64")
        self.ctx = ctx
        self.name = name
        self.index_partition = index_partition
        self.index_space = index_partition.space
        self.dtype = dtype
        self.colors = index_partition.colors_of_rank(ctx.rank, ctx.world_size)
        self.piece_sizes: List[int] = [index_partition.pieces[c].volume for c in self.colors]
        self.piece_offsets: List[int] = []
        acc = 0
        for n in self.piece_sizes:
            self.piece_offsets.append(acc)
            acc += n
        self.local_size = acc
        self.pad_lo, self.pad_hi = int(pad[0]), int(pad[1])
        if self.pad_lo < 0 or self.pad_hi < 0:
            raise ValueError("negative halo padding")
        if self.index_space.dim == 1:
            self.own_lo, self.own_hi = index_partition.rank_interval(ctx.rank, ctx.world_size)
        else:
            if self.pad_lo or self.pad_hi:
                raise ValueError("halo padding exists for 1-D vectors only")
            self.own_lo, self.own_hi = 0, -1
        self.data = ctx.backend.empty(self.pad_lo + self.local_size + self.pad_hi, dtype)
        self.owned = self.data[self.pad_lo:self.pad_lo + self.local_size]

    # -- accessors (src/PartitionedVector.hpp:52-72) ------------------------------------
    def get_name(self) -> str:
        return self.name

    def get_index_space(self):
        return self.index_space

    def get_dim(self) -> int:
        return self.index_space.dim

    def get_color_space(self) -> int:
        return self.index_partition.num_colors

    def get_index_partition(self) -> IndexPartition:
        return self.index_partition

    def piece(self, i: int) -> torch.Tensor:
        """View of the i-th local piece (colour self.colors[i])."""
        off = self.piece_offsets[i]
        return self.owned[off:off + self.piece_sizes[i]]

    def _check_same_shape(self, x: "PartitionedVector") -> None:
        # assert(index_space == x.get_index_space()) ... (src/PartitionedVector.cpp:214-216)
        if self.index_space != x.index_space or self.index_partition != x.index_partition:
            raise ValueError(f"vectors {self.name} and {x.name} live on different partitions")
        if self.dtype != x.dtype:
            raise TypeError(f"vectors {self.name} and {x.name} have different entry types")

    def _scalars(self, scalars) -> list:
        out = []
        for s in scalars:
            if not isinstance(s, Scalar):
                s = Scalar(self.ctx, s, self.dtype)
            elif s.dtype != self.dtype:
                raise TypeError("scalar and vector entry types differ")
            out.append(s.get_future())
        return out

    # -- fills and copies (src/PartitionedVector.cpp:151-191) -----------------------------
    def constant_fill(self, value) -> None:
        if isinstance(value, Scalar):
            value = value.get_value()
        for i in range(len(self.piece_sizes)):
            if self.piece_sizes[i]:
                self.ctx.backend.fill(self.piece(i), value)

    def zero_fill(self) -> None:
        self.constant_fill(0.0)

    def assign(self, x: "PartitionedVector") -> "PartitionedVector":
        """operator=(const PartitionedVector &): index copy, piece by piece."""
        self._check_same_shape(x)
        for i in range(len(self.piece_sizes)):
            if self.piece_sizes[i]:
                self.ctx.backend.copy(self.piece(i), x.piece(i))
        return x

    # -- BLAS-1 (src/PartitionedVector.cpp:195-358) ---------------------------------------
    def scal(self, alpha) -> None:
        futs = self._scalars([alpha])
        for i in range(len(self.piece_sizes)):
            if self.piece_sizes[i]:
                self.ctx.backend.scal(self.piece(i), futs)

    def axpy(self, *args) -> None:
        """axpy(alpha, x) | axpy(numer, denom, x) | axpy(numer1, numer2, denom, x):
        self <- fma(get_alpha(scalars), x, self)."""
        *scalars, x = args
        if not 1 <= len(scalars) <= 3:
            raise TypeError("axpy takes 1 to 3 scalars and a vector")
        self._check_same_shape(x)
        futs = self._scalars(scalars)
        for i in range(len(self.piece_sizes)):
            if self.piece_sizes[i]:
                self.ctx.backend.axpy(self.piece(i), futs, x.piece(i))

    def xpay(self, *args) -> None:
        """xpay(alpha, x) | xpay(numer, denom, x): self <- fma(get_alpha(scalars), self, x)."""
        *scalars, x = args
        if not 1 <= len(scalars) <= 2:
            raise TypeError("xpay takes 1 or 2 scalars and a vector")
        self._check_same_shape(x)
        futs = self._scalars(scalars)
        for i in range(len(self.piece_sizes)):
            if self.piece_sizes[i]:
                self.ctx.backend.xpay(self.piece(i), futs, x.piece(i))

    def dot_local(self, x: "PartitionedVector", out: torch.Tensor) -> None:
        """out[0] <- this rank's share of the dot product: its pieces' results folded in
        colour order.  No communication."""
        self._check_same_shape(x)
        be = self.ctx.backend
        n = len(self.piece_sizes)
        if n == 1:
            be.dot(self.piece(0), x.piece(0), out)
        else:
            slots = be.empty(n, self.dtype)
            for i in range(n):
                be.dot(self.piece(i), x.piece(i), slots[i:i + 1])
            be.sum_scalars(slots, out)

    def dot(self, x: "PartitionedVector") -> Scalar:
        """Per-piece dot products folded in colour order, then summed over ranks: the
        REDOP_SUM future-map reduction of src/PartitionedVector.cpp:353-357."""
        out = self.ctx.backend.empty(1, self.dtype)
        self.dot_local(x, out)
        self.ctx.comm.all_reduce_sum(out)
        return Scalar(self.ctx, out)

    # -- test / debug helpers ---------------------------------------------------------------
    def to_host(self) -> torch.Tensor:
        """This rank's owned elements, in colour order, on the host."""
        return self.owned.detach().cpu()
