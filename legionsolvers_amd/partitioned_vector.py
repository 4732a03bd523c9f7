"""PartitionedVector: the reference's user-facing vector (src/PartitionedVector.{hpp,cpp}).

A vector is an index space coloured by a disjoint, complete partition; every BLAS-1 method is
one leaf launch per piece (the reference's IndexTaskLauncher over the colour space,
src/PartitionedVector.cpp:195-358).  One process holds the pieces of its rank in ONE
contiguous HBM buffer, in colour order, optionally with room before and after them for the
halo of a 1-D vector, so that an SpMV reads x and its ghost cells through one base pointer
and neighbouring pieces of one rank need no copy at all.
"""
from __future__ import annotations

import weakref
from typing import List, Tuple

import torch

from .index_space import IndexPartition
from .scalar import Scalar


class _DeferredAxpy:
    """z <- fma(alpha, p, z) that has been asked for but not launched yet (see
    PartitionedVector.axpy(defer=True)); referenced from both vectors until it runs."""

    def __init__(self, z: "PartitionedVector", p: "PartitionedVector", scalars, on_outcome=None):
        self.z, self.p, self.scalars, self.on_outcome = z, p, scalars, on_outcome

    def detach(self, fused: bool) -> None:
        self.z._deferred = self.p._deferred = None
        if self.on_outcome is not None:
            self.on_outcome(fused)

    def run(self) -> None:
        """Launch it on its own, as if it had never been held back."""
        self.detach(False)
        self.z.axpy(*self.scalars, self.p)


class PartitionedVector:
    """Views of the storage (`data`, `owned`, `piece(i)`) are to be fetched again after an operation
    on the vector rather than kept across it, as a Legion accessor is valid for one mapping only:
    fetching one makes an update that is being held back (axpy(defer=True)) happen, and, because the
    views are writable, counts as a write: whatever a fused leaf had left behind about the vector's
    contents (partial sums of a dot product) is forgotten, so a write through the view can never be
    followed by a stale dot.  The package's own leaf calls use the `_own` / `_buf` / `_piece`
    forms, which settle but do not count as writes."""

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
        self._data = ctx.backend.empty(self.pad_lo + self.local_size + self.pad_hi, dtype)
        self._owned = self._data[self.pad_lo:self.pad_lo + self.local_size]
        self._deferred = None  # a _DeferredAxpy this vector takes part in
        # Bookkeeping of the fused leaf forms: `version` counts the writes made through this
        # object's methods; `_ready` describes partial sums of a dot product of this vector's
        # CURRENT contents with `partner` (possibly itself) that a fused kernel left behind.
        self.version = 0
        self._ready = None
        self._partial_bufs = {}

    # -- storage ---------------------------------------------------------------------------
    def _settle(self) -> None:
        """Run an update that was being held back, before anyone looks at or writes the storage."""
        d = self._deferred
        if d is not None:
            d.run()

    @property
    def data(self) -> torch.Tensor:
        """The whole buffer, halo room included (writable: see the class note)."""
        self._settle()
        self._touch()
        return self._data

    @property
    def owned(self) -> torch.Tensor:
        """This rank's elements, pieces in colour order (writable: see the class note)."""
        self._settle()
        self._touch()
        return self._owned

    def mark_written(self) -> None:
        """For a caller that kept a view and wrote through it later: the vector's contents changed."""
        self._settle()
        self._touch()

    # the same views for the package's own leaf calls, which do their own bookkeeping
    @property
    def _buf(self) -> torch.Tensor:
        self._settle()
        return self._data

    @property
    def _own(self) -> torch.Tensor:
        self._settle()
        return self._owned

    def _piece(self, i: int) -> torch.Tensor:
        off = self.piece_offsets[i]
        return self._own[off:off + self.piece_sizes[i]]

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
        """View of the i-th local piece (colour self.colors[i]); writable: see the class note."""
        off = self.piece_offsets[i]
        return self.owned[off:off + self.piece_sizes[i]]

    def _touch(self) -> None:
        """This vector is about to be (or was just) written: forget what was known about it."""
        self.version += 1
        self._ready = None

    def _partials(self, kind: str, i: int, slots: int) -> torch.Tensor:
        buf = self._partial_bufs.get((kind, i))
        if buf is None or buf.numel() < slots:
            buf = self.ctx.backend.empty(max(slots, 1), self.dtype)
            self._partial_bufs[(kind, i)] = buf
        return buf

    def _note_ready(self, partner: "PartitionedVector", pieces) -> None:
        """pieces[i] = (partials tensor or None, count): shares of sum over piece i of
        self * partner, valid for the current contents of both."""
        self._ready = (weakref.ref(partner), partner.version, pieces)  # weak: partner may be self

    def _ready_for(self, other: "PartitionedVector"):
        r = self._ready
        if r is not None and r[0]() is other and other.version == r[1]:
            return r[2]
        return None

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
        self._touch()
        for i in range(len(self.piece_sizes)):
            if self.piece_sizes[i]:
                self.ctx.backend.fill(self._piece(i), value)

    def zero_fill(self) -> None:
        self.constant_fill(0.0)

    def random_fill(self, seed: int = 0) -> None:
        """Uniform [0, 1) entries that depend on `seed` and the global element index only (so the
        same on any partition and any number of ranks).  The reference has the task but never
        wired it to the class (RandomFillTask, src/UtilityTasks.cpp:128-166; "TODO: random_fill"
        in test/Test02VectorOperations.cpp:128), and seeds it from std::random_device."""
        if self.index_space.dim != 1:
            raise NotImplementedError("random_fill numbers the elements of 1-D vectors")
        self._touch()
        for i, c in enumerate(self.colors):
            if self.piece_sizes[i]:
                self.ctx.backend.random_fill(self._piece(i), seed, self.index_partition.pieces[c].lo[0])

    def assign(self, x: "PartitionedVector") -> "PartitionedVector":
        """operator=(const PartitionedVector &): index copy, piece by piece."""
        self._check_same_shape(x)
        self._touch()
        for i in range(len(self.piece_sizes)):
            if self.piece_sizes[i]:
                self.ctx.backend.copy(self._piece(i), x._piece(i))
        return x

    # -- BLAS-1 (src/PartitionedVector.cpp:195-358) ---------------------------------------
    def scal(self, alpha) -> None:
        futs = self._scalars([alpha])
        self._touch()
        for i in range(len(self.piece_sizes)):
            if self.piece_sizes[i]:
                self.ctx.backend.scal(self._piece(i), futs)

    def axpy(self, *args, speculate: bool = False, defer: bool = False, on_outcome=None) -> None:
        """axpy(alpha, x) | axpy(numer, denom, x) | axpy(numer1, numer2, denom, x):
        self <- fma(get_alpha(scalars), x, self).

        speculate: use the fused axpy + squared-norm leaf, so that a dot(self, self) that
        follows before the next write only has to add up the partial sums.
        defer: hold the update back until either vector's storage is next looked at.  If the
        next thing to happen to x is x.xpay(beta, r) (CG: x_sol += alpha p, then p <- r + beta p),
        both run as one pass over x; if it is a dot product with self as an operand (CG: r -= alpha q,
        then r.r; GMRES: w -= h v_k, then v_{k+1}.w), the update leaves that dot's partial sums
        behind; anything else makes the update run first, on its own.
        Results are the same either way.  on_outcome(fused) is told which of the two happened."""
        *scalars, x = args
        if not 1 <= len(scalars) <= 3:
            raise TypeError("axpy takes 1 to 3 scalars and a vector")
        self._check_same_shape(x)
        if defer and x is not self and getattr(self.ctx.backend, "axpy_xpay", None) is not None \
                and getattr(self.ctx.backend, "axpy_dot", None) is not None:
            self._scalars(scalars)  # type checks now, not when it runs
            self._settle()
            x._settle()
            self._deferred = x._deferred = _DeferredAxpy(self, x, list(scalars), on_outcome)
            return
        futs = self._scalars(scalars)
        be = self.ctx.backend
        fused = speculate and x is not self and getattr(be, "axpy_sqnorm", None) is not None
        self._touch()
        pieces = []
        for i in range(len(self.piece_sizes)):
            if not self.piece_sizes[i]:
                pieces.append((None, 0))
            elif fused:
                buf = self._partials("sq", i, be.sqnorm_slots(self._piece(i)))
                pieces.append((buf, be.axpy_sqnorm(self._piece(i), futs, x._piece(i), buf)))
            else:
                be.axpy(self._piece(i), futs, x._piece(i))
        if fused:
            self._note_ready(self, pieces)

    def xpay(self, *args) -> None:
        """xpay(alpha, x) | xpay(numer, denom, x): self <- fma(get_alpha(scalars), self, x)."""
        *scalars, x = args
        if not 1 <= len(scalars) <= 2:
            raise TypeError("xpay takes 1 or 2 scalars and a vector")
        self._check_same_shape(x)
        futs = self._scalars(scalars)
        d = self._deferred
        if d is not None and d.p is self and x is not d.z and x is not self and x._deferred is None:
            # z <- fma(alpha, self, z) was held back: one pass over self serves both updates
            d.detach(True)
            z = d.z
            futs_a = z._scalars(d.scalars)
            self._touch()
            z._touch()
            for i in range(len(self.piece_sizes)):
                if self.piece_sizes[i]:
                    self.ctx.backend.axpy_xpay(z._piece(i), futs_a, self._piece(i), futs, x._piece(i))
            return
        self._touch()
        for i in range(len(self.piece_sizes)):
            if self.piece_sizes[i]:
                self.ctx.backend.xpay(self._piece(i), futs, x._piece(i))

    def dot_local(self, x: "PartitionedVector", out: torch.Tensor) -> None:
        """out[0] <- this rank's share of the dot product: its pieces' results folded in
        colour order.  No communication."""
        self._check_same_shape(x)
        be = self.ctx.backend
        n = len(self.piece_sizes)
        # An update of one of the operands that is being held back runs now, fused with this dot.
        d = self._deferred if self._deferred is not None and self._deferred.z is self else x._deferred
        if d is not None and d.z in (self, x):
            z = d.z
            other = x if z is self else self
            if other is z or other._deferred is None or other._deferred is d:
                d.detach(True)
                futs = z._scalars(d.scalars)
                z._touch()
                slots = out if n == 1 else be.empty(n, self.dtype)
                for i in range(n):
                    dst = slots[i:i + 1]
                    if not z.piece_sizes[i]:
                        be.sum_partials(0, None, dst)
                        continue
                    buf = z._partials("ad", i, be.sqnorm_slots(z._piece(i)))
                    count = be.axpy_dot(z._piece(i), futs, d.p._piece(i), None if other is z else other._piece(i), buf)
                    be.sum_partials(count, buf, dst)
                if n > 1:
                    be.sum_scalars(slots, out)
                return
        ready = self._ready_for(x) or x._ready_for(self)  # partial sums a fused leaf left behind

        def piece_dot(i, dst):
            if ready is not None:
                be.sum_partials(ready[i][1], ready[i][0], dst)
            else:
                be.dot(self._piece(i), x._piece(i), dst)

        if n == 1:
            piece_dot(0, out)
        else:
            slots = be.empty(n, self.dtype)
            for i in range(n):
                piece_dot(i, slots[i:i + 1])
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
        return self._own.detach().cpu()
