"""SquarePlanner: the reference's multi-space square-system planner (src/SquarePlanner.hpp).

Vector index 0 is the solution, 1 the right-hand side, 2.. the workspace the solver asks
for; every method applies one vector operation to all spaces.  On top of the reference's
bookkeeping it does the one thing Legion does implicitly there: before a matvec it brings the
ghost cells of x into the consuming GPU's memory (the READ_ONLY ghost-partition requirement of
src/CSRMatrix.cpp:202-211), here with point-to-point RCCL transfers straight into the halo
room of the source vector.
"""
from __future__ import annotations

import os
import time
from typing import List, Tuple

import torch

from .comm import HaloPlan, build_halo_plan
from .index_space import IndexPartition, IndexSpace
from .matrices import AbstractMatrix, COOMatrix, CSRMatrix, PiecePartition
from .partitioned_vector import PartitionedVector
from .scalar import Scalar


def _hull(intervals) -> Tuple[int, int]:
    live = [iv for iv in intervals if iv[1] >= iv[0]]
    if not live:
        return (0, -1)
    return min(iv[0] for iv in live), max(iv[1] for iv in live)


class _MatrixEntry:
    def __init__(self, matrix, domain_index, range_index, kernel_partition, ghost_partition, plan,
                 split=None):
        self.matrix = matrix
        self.domain_index = domain_index
        self.range_index = range_index
        self.kernel_partition: PiecePartition = kernel_partition
        self.ghost_partition: PiecePartition = ghost_partition
        self.plan: HaloPlan = plan
        self.split = split   # CSRMatrix.halo_split: rows that can be multiplied before the halo lands
        self.staging = None  # ghost buffer for source vectors without halo room
        self.rstaging = None  # rmatvec: the product over the whole window, and one landing buffer per peer


class SquarePlanner:
    def __init__(self, ctx, dtype: torch.dtype = torch.float64):
        self.ctx = ctx
        self.dtype = dtype
        self.canonical_index_spaces: List[IndexSpace] = []
        self.canonical_index_partitions: List[IndexPartition] = []
        self.sol_vectors: List[PartitionedVector] = []
        self.rhs_vectors: List[PartitionedVector] = []
        self.workspace_vectors: List[List[PartitionedVector]] = []
        self.row_partitioned_matrices: List[_MatrixEntry] = []
        # bets of the fused product + dot (see _fused_dot_pays)
        self._fused_dot_issued = 0
        self._fused_dot_used = 0
        self._paired_products = 0   # products of two systems that shared one pass over a matrix (matvec)
        self._fused_dot_pending = None
        # bets of the held-back solution update (see axpy)
        self._deferred_issued = [0, 0]  # [into the solution, into a workspace vector]
        self._deferred_fused = [0, 0]

    def get_context(self):
        return self.ctx

    # -- registration (src/SquarePlanner.hpp:99-235) --------------------------------------
    def _register(self, v: PartitionedVector, vectors: list) -> int:
        if self.workspace_vectors:
            raise RuntimeError("vectors must be added before the workspace is allocated")
        if v.dtype != self.dtype:
            raise TypeError("vector entry type differs from the planner's")
        idx = len(vectors)
        if len(self.canonical_index_spaces) > idx:
            if self.canonical_index_spaces[idx] != v.get_index_space():
                raise ValueError("index space differs from the canonical one of this slot")
            if self.canonical_index_partitions[idx] != v.get_index_partition():
                raise ValueError("partition differs from the canonical one of this slot")
        else:
            self.canonical_index_spaces.append(v.get_index_space())
            self.canonical_index_partitions.append(v.get_index_partition())
        vectors.append(v)
        return idx

    def add_sol_vector(self, v: PartitionedVector) -> int:
        return self._register(v, self.sol_vectors)

    def add_rhs_vector(self, v: PartitionedVector) -> int:
        return self._register(v, self.rhs_vectors)

    def get_num_spaces(self) -> int:
        n = len(self.canonical_index_spaces)
        if not (n == len(self.sol_vectors) == len(self.rhs_vectors)):
            raise RuntimeError("every space needs one solution and one right-hand-side vector")
        return n

    def add_row_partitioned_matrix(self, matrix: AbstractMatrix, domain_index: int,
                                   range_index: int) -> None:
        n = self.get_num_spaces()
        if not (0 <= domain_index < n and 0 <= range_index < n):
            raise IndexError("matrix refers to a space the planner does not have")
        if matrix.dtype != self.dtype:
            raise TypeError("matrix entry type differs from the planner's")
        range_partition = self.canonical_index_partitions[range_index]
        domain_partition = self.canonical_index_partitions[domain_index]
        domain_space = self.canonical_index_spaces[domain_index]
        if domain_space.dim != 1 or self.canonical_index_spaces[range_index].dim != 1:
            raise ValueError("matrices act on 1-D index spaces")
        kernel_partition = matrix.create_kernel_partition_from_range_partition(range_partition)
        ghost_partition = matrix.create_domain_partition_from_kernel_partition(
            domain_space, kernel_partition)
        # Who owns and who reads which interval of x: gathered once, then fixed.
        rank, world = self.ctx.rank, self.ctx.world_size
        mine = (domain_partition.rank_interval(rank, world), _hull(ghost_partition.intervals))
        everyone = self.ctx.comm.all_gather_object(mine)
        plan = build_halo_plan(rank, [e[0] for e in everyone], [e[1] for e in everyone])
        # The window of x every product of this matrix will be given (see _gather_ghost).
        own = domain_partition.rank_interval(rank, world)
        window = (own[0] - plan.pad_lo, own[1] + plan.pad_hi) if (plan.recvs or plan.sends) else own
        # Set-up-time copies, the best first: the sliced-pattern copy (CSR matrices whose rows repeat a few column
        # patterns), else the 32-bit copy of the structure.  A matrix that got the first does not get the second
        # as well: its products would never read it, and it is 4 bytes per entry of HBM (a sixth of what a CSR
        # matrix of fp64 entries takes with its sliced copy) that a larger problem can use.
        slicer = getattr(matrix, "slice_structure", None)
        if slicer is not None and not getattr(matrix, "_slice_tried", False):
            slicer(window[0], window[1], kernel_partition, range_partition)
            matrix._slice_tried = True
        compress = getattr(matrix, "compress", None)
        if compress is not None and getattr(matrix, "col32", None) is None and not getattr(matrix, "sliced", None):
            compress(window[0], window[1], kernel_partition, range_partition)
        tune = getattr(matrix, "tune", None)
        if tune is not None and not getattr(matrix, "_tuned", False) and not getattr(matrix, "sliced", None):
            tune(window[0], window[1], kernel_partition, range_partition)
            matrix._tuned = True
        split = None
        if plan.recvs and getattr(matrix, "halo_split", None) is not None \
                and os.environ.get("LSK_OVERLAP", "1") != "0":
            own_lo, own_hi = domain_partition.rank_interval(rank, world)
            split = matrix.halo_split(own_lo, own_hi, kernel_partition, range_partition)
            if split is not None and not any(run[0] == 0 for runs in split.values() for run in runs):
                split = None  # every row waits for the halo anyway
        self.row_partitioned_matrices.append(
            _MatrixEntry(matrix, domain_index, range_index, kernel_partition, ghost_partition, plan, split))

    def _halo_pad(self, space_index: int) -> Tuple[int, int]:
        lo = hi = 0
        for e in self.row_partitioned_matrices:
            if e.domain_index == space_index:
                lo, hi = max(lo, e.plan.pad_lo), max(hi, e.plan.pad_hi)
        return lo, hi

    def allocate_workspace(self, num_vectors: int) -> None:
        """Workspace vectors get halo room for every matrix that reads their space, so a
        matvec gathers ghost cells in place (src/SquarePlanner.hpp:153-188 allocates plain
        fields; the ghost instances are Legion's)."""
        if self.workspace_vectors:
            raise RuntimeError("workspace already allocated")
        n = self.get_num_spaces()
        need = num_vectors * sum((self.canonical_index_partitions[i].space.volume + sum(self._halo_pad(i))) * self.dtype.itemsize
                                 for i in range(n))
        hold = self._room_in_front_of_the_workspace(need)
        backend = self.ctx.backend

        def let_go() -> bool:
            if not hold:
                return False
            del hold[:]
            torch.cuda.empty_cache()
            backend.optional_memory_stays = False
            return True

        # While blocks are held in front of the workspace, a vector that finds no room must not cost a matrix its
        # set-up-time copy (HipBackend.empty): the blocks go instead, and the vector is tried again.
        backend.optional_memory_stays = bool(hold)
        try:
            for j in range(num_vectors):
                row = []
                for i in range(n):
                    while True:
                        try:
                            row.append(PartitionedVector(self.ctx, f"workspace_{j}_{i}",
                                                         self.canonical_index_partitions[i], self.dtype,
                                                         pad=self._halo_pad(i)))
                            break
                        except torch.cuda.OutOfMemoryError:
                            if not let_go():
                                raise
                            self.workspace_placement["blocks_in_front"] = 0   # (the rest lies where it would have anyway)
                self.workspace_vectors.append(row)
        finally:
            let_go()

    PLACE_MIN_GAIN = 0.015   # a deeper place for the workspace must beat the nearest by this much to be taken
    PLACE_MIN_BYTES = 1 << 30   # ... and only products that stream at least this much from HBM are placed by (a smaller
    #                             copy lives in the caches, where its product's time says nothing about HBM)

    @staticmethod
    def _choose_place(times, min_gain: float) -> int:
        """Index of the place to take, given the product's time at each (nearest first): every step deeper must be
        faster than the place chosen so far by min_gain; ties and small gains go to the nearer place."""
        best = 0
        for depth in range(1, len(times)):
            if times[depth] < (1.0 - min_gain) * times[best]:
                best = depth
        return best

    def _room_in_front_of_the_workspace(self, workspace_bytes: int = 0) -> list:
        """Where in HBM the workspace vectors go.  The 288 GB of an MI355X answer in three parts of 96 GB (the three
        ranks of its 12-high stacks): two read streams are fastest inside one part (a dot product of two 8 GiB
        buffers: 7.1 against 6.7 TB/s), a read and a write stream in two different ones (a copy: 6.2 against 5.7
        TB/s; tools/exp/hbm_pair_rates.py).  A product reads the matrix values and writes a workspace vector, so
        the workspace is best kept out of the part the values of a large sliced-pattern copy lie in: CG on the
        27-point 512^3 matrix 161 -> 171 it/s (tools/exp/rank_region_placement.py).  No interface says where an
        allocation lands; on these devices memory is handed out from the bottom up, so a block of a third of the
        device allocated first pushes what follows into the next part.  Nothing is assumed, though: one product
        of every such matrix is timed into a vector here, into one behind such a block and into one behind two,
        and the workspace goes one place deeper for as long as that is faster by PLACE_MIN_GAIN (ties go to the
        nearer place; else it stays where it would have gone anyway).  Returns the blocks to hold while the
        workspace is allocated (fewer when workspace_bytes would not fit behind them; the caller frees them right
        after; they are never touched).  Skipped without a sliced copy of at least PLACE_MIN_BYTES of values, with
        LSK_TUNE=0 or LSK_PLACE_VECTORS=0, and whenever the device has no room for the blocks; the times are kept
        in self.workspace_placement."""
        self.workspace_placement = None
        dev = self.ctx.device
        if (dev.type != "cuda" or os.environ.get("LSK_TUNE", "1") == "0" or os.environ.get("LSK_PLACE_VECTORS", "1") == "0"
                or torch.cuda.is_current_stream_capturing() or self.ctx.comm.backend == "gloo"):
            return []   # (gloo: a rehearsal with several ranks on one card, whose blocks would take each other's memory)
        targets = []   # the first large sliced piece of every matrix: the sum of their products is what is compared
        for e in self.row_partitioned_matrices:
            m = e.matrix
            if not getattr(m, "sliced", None) or getattr(m, "sliced_window", None) is None or not hasattr(m, "time_piece"):
                continue
            rp = self.canonical_index_partitions[e.range_index]
            for i, c in enumerate(e.kernel_partition.colors):
                if (c in m.sliced and rp.pieces[c].volume >= m.TUNE_MIN_ROWS
                        and m.sliced[c].padded_entries * self.dtype.itemsize >= self.PLACE_MIN_BYTES):
                    targets.append((m, i, c, rp.pieces[c].volume, m.sliced_window[1] - m.sliced_window[0] + 1))
                    break
        if not targets:
            return []
        rows, window = max(t[3] for t in targets), max(t[4] for t in targets)
        began = time.perf_counter()
        third = torch.cuda.mem_get_info(dev)[1] // 3
        # kept: input 0, output 0, block 1, input 1, output 1, block 2, input 2, output 2 -- at every place the product
        # runs from a first vector into a second one, as the solver's products will from the first workspace vector
        # into the second (what is freed here is what the workspace is then allocated from)
        kept, times = [], []
        try:
            for depth in range(3):
                if depth:
                    if torch.cuda.mem_get_info(dev)[0] < third + (4 << 30):
                        break
                    kept.append(torch.empty(third, dtype=torch.uint8, device=dev))
                kept.append(torch.zeros(window, dtype=self.dtype, device=dev))
                kept.append(torch.empty(rows, dtype=self.dtype, device=dev))
                times.append(sum(m.time_piece(i, c, kept[-2][:w], kept[-1][:r]) for m, i, c, r, w in targets))
        except torch.cuda.OutOfMemoryError:
            while kept and (len(kept) < 2 or (len(kept) - 2) % 3):   # a place without all of its vectors
                kept.pop()
        if not times:
            return []
        best = self._choose_place(times, self.PLACE_MIN_GAIN)
        hold = kept[:3 * best]   # everything in front of the chosen place stays until the workspace is there
        del kept
        torch.cuda.synchronize(dev)
        torch.cuda.empty_cache()   # (what was freed goes back to the device, not into torch's cache, which would hand
        #                             a vector's block from behind the chosen place to the first workspace vector)
        while hold and torch.cuda.mem_get_info(dev)[0] < workspace_bytes + (2 << 30):
            del hold[-3:]          # the workspace itself must fit behind what is held: else the deepest block goes
            torch.cuda.empty_cache()
            best -= 1
        self.workspace_placement = {"blocks_in_front": best, "product_ms_by_blocks_in_front": [round(t, 4) for t in times],
                                    "seconds": round(time.perf_counter() - began, 3)}
        return hold

    def get_vector(self, vec_idx: int, space_idx: int) -> PartitionedVector:
        if vec_idx == 0:
            return self.sol_vectors[space_idx]
        if vec_idx == 1:
            return self.rhs_vectors[space_idx]
        return self.workspace_vectors[vec_idx - 2][space_idx]

    def _scalar(self, s) -> Scalar:
        return s if isinstance(s, Scalar) else Scalar(self.ctx, s, self.dtype)

    def settle(self) -> None:
        """Brings every vector's storage up to date and forgets what the fused leaves left for a call that
        may follow (held-back updates run on their own; partial sums of dot products and the dot shares
        of the last product are dropped): after it, what the next operation launches depends on nothing
        an earlier one left behind.  The boundary a recorded step needs (solvers.CapturedStep); values
        are unchanged, later dots are merely computed afresh."""
        for v in self.sol_vectors + self.rhs_vectors + [w for row in self.workspace_vectors for w in row]:
            v._settle()
            v._ready = None
        self._fused_dot_pending = None

    # -- vector operations over all spaces (src/SquarePlanner.hpp:248-338) -----------------
    def zero_fill(self, vec_idx: int) -> None:
        for i in range(self.get_num_spaces()):
            self.get_vector(vec_idx, i).zero_fill()

    def copy(self, dst_index: int, src_index: int) -> None:
        for i in range(self.get_num_spaces()):
            self.get_vector(dst_index, i).assign(self.get_vector(src_index, i))

    def scal(self, dst_index: int, alpha) -> None:
        alpha = self._scalar(alpha)
        for i in range(self.get_num_spaces()):
            self.get_vector(dst_index, i).scal(alpha)

    def axpy(self, dst_index: int, *args) -> None:
        """axpy(dst, alpha, src) | axpy(dst, numer, denom, src) | axpy(dst, n1, n2, denom, src)."""
        *scalars, src_index = args
        scalars = [self._scalar(s) for s in scalars]
        # Workspace vectors are only ever written through the planner, so the fused
        # axpy + squared norm can serve a dot(dst, dst) that follows (CG: r.r after r -= a q).
        # Held-back updates (PartitionedVector.axpy(defer=True)): an update of the solution by a
        # workspace vector may share one pass with that vector's own update (CG: x += alpha p ...
        # p <- r + beta p); an update of a workspace vector may leave the partial sums of the dot
        # product that follows (CG: r -= alpha q, r.r; GMRES: w -= h v_k, v_{k+1}.w).  Bets, like the
        # fused product + dot: each kind is kept up only while at least half of them are won.
        kind = 0 if dst_index == 0 else 1
        issued, won = self._deferred_issued[kind], self._deferred_fused[kind]
        # (this record does not fade as that of the fused dots does, _fused_dot_pays: a held-back update that
        # finds no partner is settled by a pass of its own, and BiCGStab, which loses two of these bets in
        # three, went from 177 to 143 it/s on the random COO matrix when it was offered them again and again)
        defer = (self.ctx.fuse and src_index >= 2 and src_index != dst_index and dst_index != 1
                 and (issued < 4 or 2 * won >= issued))
        # (the older form of the second bet, for when it is not being placed: axpy + squared norm)
        speculate = self.ctx.fuse and dst_index >= 2 and not defer
        outcome = (lambda fused, k=kind: self._deferred_outcome(k, fused)) if defer else None
        for i in range(self.get_num_spaces()):
            self.get_vector(dst_index, i).axpy(*scalars, self.get_vector(src_index, i), speculate=speculate,
                                               defer=defer, on_outcome=outcome)

    def _deferred_outcome(self, kind: int, fused: bool) -> None:
        self._deferred_issued[kind] += 1
        self._deferred_fused[kind] += 1 if fused else 0

    def xpay(self, dst_index: int, *args) -> None:
        """xpay(dst, alpha, src) | xpay(dst, numer, denom, src)."""
        *scalars, src_index = args
        scalars = [self._scalar(s) for s in scalars]
        for i in range(self.get_num_spaces()):
            self.get_vector(dst_index, i).xpay(*scalars, self.get_vector(src_index, i))

    def dot(self, v_idx: int, w_idx: int) -> Scalar:
        """Sum over spaces of the per-space dot products (src/SquarePlanner.hpp:331-338).  The
        per-space results cross the ranks in ONE all-reduce of num_spaces scalars and are then
        added left to right, as the reference's chain of AddScalarTasks does."""
        n = self.get_num_spaces()
        be = self.ctx.backend
        if self._fused_dot_pending is not None and self._fused_dot_pending == frozenset((v_idx, w_idx)):
            self._fused_dot_used += 1
        self._fused_dot_pending = None
        slots = be.empty(n, self.dtype)
        for i in range(n):
            self.get_vector(v_idx, i).dot_local(self.get_vector(w_idx, i), slots[i:i + 1])
        self.ctx.comm.all_reduce_sum(slots)
        if n == 1:
            return Scalar(self.ctx, slots)
        out = be.empty(1, self.dtype)
        be.sum_scalars(slots, out)
        return Scalar(self.ctx, out)

    def multi_dot(self, v_indices, w_idx: int) -> torch.Tensor:
        """k dot products with one vector in one pass over it: out[j] = dot(v_indices[j], w_idx), the
        sum over spaces and ranks included (one all-reduce of k scalars).  Returns the k device
        scalars as one tensor; Scalar(ctx, out[j:j+1]) wraps one.  The reference's planner has no such
        method (its GMRES takes its inner products one by one, src/GMRESSolver.hpp:92-100)."""
        be = self.ctx.backend
        k, n = len(v_indices), self.get_num_spaces()
        if not 1 <= k <= be.MULTI_MAX:
            raise ValueError(f"multi_dot takes 1..{be.MULTI_MAX} vectors")
        self._fused_dot_pending = None
        def fold(rows):
            """Row of k sums over the given rows of k values, added left to right."""
            if len(rows) == 1:
                return rows[0]
            stacked = torch.stack(rows, dim=1).contiguous()   # (k, len(rows))
            folded = be.empty(k, self.dtype)
            for j in range(k):
                be.sum_scalars(stacked[j], folded[j:j + 1])
            return folded

        per_space = []  # as dot(): the pieces of a space folded in colour order ...
        for i in range(n):
            w = self.get_vector(w_idx, i)
            vs = [self.get_vector(v, i) for v in v_indices]
            for v in vs:
                w._check_same_shape(v)
            rows = []
            for p in range(len(w.piece_sizes)):
                row = be.empty(k, self.dtype)
                be.multi_dot([v._piece(p) for v in vs], w._piece(p), row)
                rows.append(row)
            per_space.append(fold(rows) if rows else torch.zeros(k, dtype=self.dtype, device=self.ctx.device))
        if n == 1:
            out = per_space[0]
            self.ctx.comm.all_reduce_sum(out)
            return out
        # ... the spaces' results summed over the ranks in ONE all-reduce, then added left to right
        slots = torch.stack(per_space, dim=0).contiguous()       # (spaces, k)
        self.ctx.comm.all_reduce_sum(slots)
        return fold([slots[i] for i in range(n)])

    def multi_axpy(self, dst_index: int, coef: torch.Tensor, src_indices, negate: bool = False) -> None:
        """dst <- dst + sum_j (+-coef[j]) src_j as the chain of axpys in ascending j, in one pass over dst
        per group of 8 vectors; leaves the shares of dot(dst, dst) for a dot product that follows."""
        be = self.ctx.backend
        if coef.numel() != len(src_indices):
            raise ValueError("one coefficient per source vector")
        for i in range(self.get_num_spaces()):
            dst = self.get_vector(dst_index, i)
            srcs = [self.get_vector(v, i) for v in src_indices]
            for v in srcs:
                dst._check_same_shape(v)
                if v is dst:
                    raise ValueError("a source vector is the destination")
            views = [[v._piece(p) for v in srcs] for p in range(len(dst.piece_sizes))]
            dst._settle()
            dst._touch()
            pieces = []
            for p in range(len(dst.piece_sizes)):
                if not dst.piece_sizes[p]:
                    pieces.append((None, 0))
                    continue
                buf = dst._partials("sq", p, be.sqnorm_slots(dst._piece(p)))
                pieces.append((buf, be.multi_axpy(dst._piece(p), coef, views[p], negate, buf)))
            dst._note_ready(dst, pieces)

    # -- matvec (src/SquarePlanner.hpp:340-357) ---------------------------------------------
    def _gather_ghost(self, e: _MatrixEntry, src: PartitionedVector):
        """Returns (x tensor, x_lo, x_hi, handle): the ghost instance of `src` for matrix entry
        `e`, with the transfer of the cells other ranks own posted; it has landed once
        comm.exchange_finish(handle) has been called (handle None: nothing to wait for)."""
        plan = e.plan
        own_lo, own_hi = src.own_lo, src.own_hi
        if not plan.recvs and not plan.sends:
            return src._own, own_lo, own_hi, None
        if src.pad_lo >= plan.pad_lo and src.pad_hi >= plan.pad_hi:
            buf, base = src._buf, own_lo - src.pad_lo      # gather in place
        else:
            need = plan.pad_lo + src.local_size + plan.pad_hi
            if e.staging is None or e.staging.numel() != need:
                e.staging = self.ctx.backend.empty(need, self.dtype)
            buf, base = e.staging, own_lo - plan.pad_lo
            if src.local_size:
                self.ctx.backend.copy(buf[plan.pad_lo:plan.pad_lo + src.local_size], src._own)
        sends = [(t.peer, src._own[t.start - own_lo:t.start - own_lo + t.count]) for t in plan.sends]
        recvs = [(t.peer, buf[t.start - base:t.start - base + t.count]) for t in plan.recvs]
        handle = self.ctx.comm.exchange_start(sends, recvs)
        x_lo = own_lo - plan.pad_lo
        x_hi = own_hi + plan.pad_hi
        return buf[x_lo - base:x_hi - base + 1], x_lo, x_hi, handle

    def matvec(self, dst_idx: int, src_idx: int) -> None:
        """dst <- A src over all registered matrices.  The first CSR matrix into a range space
        overwrites it (beta = 0), every further matrix accumulates (beta = 1); a COO matrix
        always accumulates, onto a zero fill if it comes first.  The result equals the
        reference's zero_fill followed by accumulating matvecs."""
        touched = set()
        writers = {}
        for e in self.row_partitioned_matrices:
            writers[e.range_index] = writers.get(e.range_index, 0) + 1
        # dot(src, dst) usually follows (CG's p.Ap): where one CSR matrix alone produces a range
        # space and both vectors are planner workspace, its product leaves that dot's partial
        # sums behind -- for short rows only, where the saved pass over two vectors outweighs
        # what the fused kernel costs (CSRMatrix.fuses_dot).
        fuse = (self.ctx.fuse and dst_idx >= 2 and src_idx >= 2 and dst_idx != src_idx
                and self._fused_dot_pays())
        fused_any = False
        paired = set()
        entries = self.row_partitioned_matrices
        for idx, e in enumerate(entries):
            if idx in paired:
                continue
            # Two systems on ONE matrix (the reference's BenchmarkStencil): both products in one pass
            # over the matrix values, where each alone produces its range space and no halo is due.
            mate = self._pair_mate(idx, writers, touched, paired)
            if mate is not None:
                e2 = entries[mate]
                dsts = [self.get_vector(dst_idx, f.range_index) for f in (e, e2)]
                srcs = [self.get_vector(src_idx, f.domain_index) for f in (e, e2)]
                with_dot = fuse and e.matrix.fuses_dot()
                dot_with = [self.get_vector(src_idx, f.range_index) for f in (e, e2)] if with_dot else None
                if self._matvec_pair(e, e2, dsts, srcs, dot_with):
                    self._paired_products += 1
                    paired.add(mate)
                    touched.update((e.range_index, e2.range_index))
                    fused_any = fused_any or with_dot
                    continue
            dst = self.get_vector(dst_idx, e.range_index)
            src = self.get_vector(src_idx, e.domain_index)
            first = e.range_index not in touched
            touched.add(e.range_index)
            x, x_lo, x_hi, handle = self._gather_ghost(e, src)
            comm = self.ctx.comm
            if isinstance(e.matrix, COOMatrix):
                overwrites = first and e.matrix.overwrites(x_lo, x_hi, e.kernel_partition, dst.index_partition)
                if (overwrites and handle is not None
                        and e.matrix.phases(x_lo, x_hi, e.kernel_partition, dst.index_partition)):
                    # an unstructured matrix reads all of x: what reads only the columns this rank owns is
                    # multiplied while the others' parts are on their way, the rest once they are here
                    own = (src.own_lo, src.own_hi)
                    e.matrix.matvec(dst, x, x_lo, x_hi, e.kernel_partition, 0, phase=0, own=own)
                    comm.exchange_finish(handle)
                    e.matrix.matvec(dst, x, x_lo, x_hi, e.kernel_partition, 0, phase=1, own=own)
                    self.coo_products_in_phases = getattr(self, "coo_products_in_phases", 0) + 1
                    continue
                comm.exchange_finish(handle)
                if overwrites:
                    e.matrix.matvec(dst, x, x_lo, x_hi, e.kernel_partition, 0)
                    continue
                if first:
                    dst.zero_fill()
                e.matrix.matvec(dst, x, x_lo, x_hi, e.kernel_partition, 1)
            elif (fuse and first and writers[e.range_index] == 1 and e.matrix.fuses_dot()
                  and not (e.split is not None and handle is not None)):
                comm.exchange_finish(handle)
                e.matrix.matvec(dst, x, x_lo, x_hi, e.kernel_partition, 0,
                                dot_with=self.get_vector(src_idx, e.range_index))
                fused_any = True
            elif e.split is not None and handle is not None:
                # rows that read no ghost cell are multiplied while the halo is in flight (and on the
                # sliced-pattern copy both phases can still leave the shares of the dot)
                beta = 0 if first else 1
                with_dot = (fuse and first and writers[e.range_index] == 1 and e.matrix.fuses_dot()
                            and bool(getattr(e.matrix, "sliced", None)))
                dot_with = self.get_vector(src_idx, e.range_index) if with_dot else None
                e.matrix.matvec(dst, x, x_lo, x_hi, e.kernel_partition, beta, split=e.split, phase=0,
                                dot_with=dot_with)
                comm.exchange_finish(handle)
                e.matrix.matvec(dst, x, x_lo, x_hi, e.kernel_partition, beta, split=e.split, phase=1,
                                dot_with=dot_with)
                fused_any = fused_any or with_dot
            else:
                comm.exchange_finish(handle)
                e.matrix.matvec(dst, x, x_lo, x_hi, e.kernel_partition, 0 if first else 1)
        for i in range(self.get_num_spaces()):
            if i not in touched:
                self.get_vector(dst_idx, i).zero_fill()
        if fused_any:
            self._fused_dot_issued += 1
            self._fused_dot_pending = frozenset((dst_idx, src_idx))
        else:
            self._fused_dot_pending = None

    def _matvec_pair(self, e, e2, dsts, srcs, dot_with) -> bool:
        """The products of entries e and e2 (mates, see _pair_mate) in one pass over the matrix values.
        Where ghost cells are due, both vectors' exchanges are posted, the rows that read none are
        multiplied while they are in flight, and the rest afterwards -- as matvec does for one vector.
        False: nothing was launched or posted."""
        comm = self.ctx.comm
        if not (e.plan.recvs or e.plan.sends):
            return e.matrix.matvec_pair(dsts, [v._own for v in srcs], srcs[0].own_lo, srcs[0].own_hi,
                                        e.kernel_partition, dot_with)
        x_lo, x_hi = srcs[0].own_lo - e.plan.pad_lo, srcs[0].own_hi + e.plan.pad_hi
        if not e.matrix.pairs(x_lo, x_hi, e.kernel_partition, dsts[0].index_partition):
            return False
        gathered = [self._gather_ghost(f, v) for f, v in ((e, srcs[0]), (e2, srcs[1]))]
        xs = [g[0] for g in gathered]
        args = (dsts, xs, x_lo, x_hi, e.kernel_partition, dot_with)
        ok = True
        if e.split is not None:
            ok = e.matrix.matvec_pair(*args, split=e.split, phase=0)
        for g in gathered:
            comm.exchange_finish(g[3])
        ok = ok and (e.matrix.matvec_pair(*args, split=e.split, phase=1) if e.split is not None
                     else e.matrix.matvec_pair(*args))
        if not ok:
            raise RuntimeError("the two-vector product declined a matrix it said it serves")
        return True

    def _pair_mate(self, idx: int, writers, touched, paired):
        """Index of a later matrix entry that can share one pass over the values with entry idx: the
        same CSR matrix object on the same partitions with the same halo plan, other spaces, each the
        only writer of its range space; None if there is none."""
        entries = self.row_partitioned_matrices
        e = entries[idx]
        if (not isinstance(e.matrix, CSRMatrix) or writers[e.range_index] != 1 or e.range_index in touched
                or e.domain_index != e.range_index):
            return None
        dst = self.get_vector(0, e.range_index)
        if not e.matrix.pairs(dst.own_lo - e.plan.pad_lo, dst.own_hi + e.plan.pad_hi, e.kernel_partition,
                              dst.index_partition):
            return None
        for j in range(idx + 1, len(entries)):
            f = entries[j]
            if (j in paired or f.matrix is not e.matrix or f.range_index == e.range_index
                    or writers[f.range_index] != 1 or f.range_index in touched
                    or f.plan.recvs != e.plan.recvs or f.plan.sends != e.plan.sends or f.split != e.split
                    or f.domain_index != f.range_index or f.kernel_partition.intervals != e.kernel_partition.intervals
                    or f.kernel_partition.colors != e.kernel_partition.colors):
                continue
            other = self.get_vector(0, f.range_index)
            if other.index_partition != dst.index_partition or other.own_lo != dst.own_lo or other.own_hi != dst.own_hi:
                continue
            return j
        return None

    def rmatvec(self, dst_idx: int, src_idx: int) -> None:
        """dst <- A^T src over all registered matrices: vector `src_idx` is read in every matrix's
        range space, vector `dst_idx` written in its domain space.  The reference registers the
        leaf tasks for this (src/Initialize.cpp:464-479) but their bodies are assert(false) and its
        planner has no such method; it is what LSQR / BiCG-type solvers need.
        Every process multiplies the transpose of its own block of rows, which yields sums over
        the window of the domain its rows touch: the part it owns is added to dst, the rest is the
        halo exchange of matvec run backwards -- sent to the owners, who add what arrives in
        ascending order of the sender's rank (deterministic)."""
        be, comm = self.ctx.backend, self.ctx.comm
        touched = set()
        for e in self.row_partitioned_matrices:
            dst = self.get_vector(dst_idx, e.domain_index)
            src = self.get_vector(src_idx, e.range_index)
            first = e.domain_index not in touched
            touched.add(e.domain_index)
            dst._touch()
            plan = e.plan
            own_lo, own_hi = dst.own_lo, dst.own_hi
            if not plan.recvs and not plan.sends:
                e.matrix.rmatvec(dst._own, own_lo, own_hi, src._own, src.own_lo, 0 if first else 1)
                continue
            w_lo, w_hi = own_lo - plan.pad_lo, own_hi + plan.pad_hi
            if e.rstaging is None:
                e.rstaging = (be.empty(w_hi - w_lo + 1, self.dtype),
                              [(t, be.empty(t.count, self.dtype)) for t in sorted(plan.sends, key=lambda t: t.peer)])
            window, landing = e.rstaging
            e.matrix.rmatvec(window, w_lo, w_hi, src._own, src.own_lo, 0)
            handle = comm.exchange_start(
                [(t.peer, window[t.start - w_lo:t.start - w_lo + t.count]) for t in plan.recvs],
                [(t.peer, buf) for t, buf in landing])
            if dst.local_size:
                mine = window[own_lo - w_lo:own_lo - w_lo + dst.local_size]
                if first:
                    be.copy(dst._own, mine)
                else:
                    be.axpy(dst._own, [], mine)   # no futures: alpha = 1
            comm.exchange_finish(handle)
            for t, buf in landing:
                be.axpy(dst._own[t.start - own_lo:t.start - own_lo + t.count], [], buf)
        for i in range(self.get_num_spaces()):
            if i not in touched:
                self.get_vector(dst_idx, i).zero_fill()
        self._fused_dot_pending = None

    def _fused_dot_pays(self) -> bool:
        """The fused product + dot is a bet on the caller's next dot product (CG wins it every
        step, GMRES once in 31 products, BiCGStab never): after a few products, keep betting
        only while at least half of the bets were won.  A product that goes without a bet makes the
        record fade a little, so that a caller who changes its ways (a run of plain products, then CG)
        is offered bets again: one in three products or so while they keep being lost (on the
        sliced-pattern copy a lost bet costs nothing: the shares' second operand is the product's own input)."""
        if self._fused_dot_issued < 4 or 2 * self._fused_dot_used >= self._fused_dot_issued:
            return True
        self._fused_dot_issued *= 0.9
        self._fused_dot_used *= 0.9
        return False
