"""Matrix classes: the reference's AbstractMatrix / CSRMatrix / COOMatrix
(src/AbstractMatrix.hpp, src/CSRMatrix.{hpp,cpp}, src/COOMatrix.{hpp,cpp}).

A matrix object holds THIS RANK's shard in the reference's own layout:
  CSR  rects  (rows, 2) int64   {lo, hi} inclusive GLOBAL kernel indices, one per owned row
       col    (nnz,)    int64   global column ids        (kernel entries k_base .. k_base+nnz-1)
       val    (nnz,)    T
  COO  row, col (nnz,) int64 global ids, val (nnz,) T.
The partition methods keep the reference's names; they return the per-piece kernel and ghost
intervals that Legion's dependent partitioning would produce (as bounding intervals), computed
on the device.  `matvec` is one leaf launch per local piece.
"""
from __future__ import annotations

import os
import sys
import time
from dataclasses import dataclass
from typing import List, Tuple

import torch

from .index_space import IndexPartition, IndexSpace
from .partitioned_vector import PartitionedVector

Interval = Tuple[int, int]


@dataclass
class PiecePartition:
    """One inclusive interval per LOCAL piece (colour order); hi < lo means empty."""
    colors: range
    intervals: List[Interval]


class AbstractMatrix:
    """Interface of src/AbstractMatrix.hpp as the planner uses it."""

    dtype: torch.dtype

    def create_kernel_partition_from_range_partition(self, range_partition: IndexPartition) -> PiecePartition:
        raise NotImplementedError

    def create_domain_partition_from_kernel_partition(self, domain_space: IndexSpace,
                                                      kernel_partition: PiecePartition) -> PiecePartition:
        raise NotImplementedError

    def matvec(self, dst: PartitionedVector, x: torch.Tensor, x_lo: int, x_hi: int,
               kernel_partition: PiecePartition, beta: int) -> None:
        raise NotImplementedError

    def rmatvec(self, out: torch.Tensor, out_lo: int, out_hi: int, src: torch.Tensor, src_lo: int,
                beta: int) -> None:
        """out <- A_local^T src + beta out, where A_local is this rank's block of rows, `src` holds
        the rank's rows (global index of src[0]: src_lo) and `out` the window out_lo..out_hi of the
        domain.  CSRRmatvecTask / COORmatvecTask are registered by the reference but their bodies
        are assert(false) (src/CSRMatrixTasks.cpp:95-100, src/COOMatrixTasks.cpp:78-83) and nothing
        launches them; here the product runs on the transpose built once at set-up."""
        raise NotImplementedError


class TransposedBlock:
    """CSR form, in the 32-bit structure, of the transpose of one rank's block of a matrix: one
    row per element of the window x_lo..x_hi of the domain (columns outside it are dropped, as
    every product drops them), column ids counted from the rank's first row.  Built by stable
    sorts of the entries by column, so inside a row of the transpose the entries keep the order
    they have in the matrix; products are deterministic (no atomics).

    The entries are transposed in runs of at most CHUNK (a sort handles fewer than 2^31, and its
    scratch stays bounded); every run becomes a structure of its own over the part of the window
    its columns span, and the runs' products accumulate in order.  For row-sorted entries of a
    banded matrix those parts barely overlap, so the extra passes over `out` are short."""

    CHUNK = 1 << 30

    def __init__(self, be, col: torch.Tensor, row32: torch.Tensor, val: torch.Tensor,
                 x_lo: int, x_hi: int, local_rows: int):
        """col: the int64 global column ids; row32: rows counted from the rank's first row."""
        self.window = (int(x_lo), int(x_hi))
        self.local_rows = int(local_rows)
        self.parts = []  # (offset into the window, rows, row offsets, column ids, values, entries)
        for k0 in range(0, val.numel() if x_hi >= x_lo else 0, self.CHUNK):
            k1 = min(k0 + self.CHUNK, val.numel())
            lo, hi = be.index_bounds(col[k0:k1])
            lo, hi = max(lo, x_lo), min(hi, x_hi)
            if hi < lo:
                continue
            extent = hi - lo + 1
            keys = be.csr_compress_cols(col[k0:k1], lo, hi)
            order = be.coo_block_order(keys, extent, 1)
            keys = be.gather(keys, order)
            rowoff = be.sorted_key_offsets(keys, extent)
            del keys
            nnz = int(rowoff[-1].item()) & 0xFFFFFFFF  # entries with a column inside the window
            if nnz:
                self.parts.append((lo - x_lo, extent, rowoff, be.gather(row32[k0:k1], order)[:nnz].clone(),
                                   be.gather(val[k0:k1], order)[:nnz].clone(), nnz))

    def apply(self, be, out: torch.Tensor, src: torch.Tensor, beta: int) -> None:
        extent = self.window[1] - self.window[0] + 1
        if out.numel() != max(extent, 0) or src.numel() != self.local_rows:
            raise ValueError("rmatvec: vectors do not match the block the transpose was built for")
        if extent <= 0:
            return
        whole = len(self.parts) == 1 and self.parts[0][1] == extent
        if beta == 0 and not whole:
            be.fill(out, 0)
        for off, rows, rowoff, col32, val, nnz in self.parts:
            be.csr_spmv(rows, nnz, rowoff, 0, col32, val, src, 0, self.local_rows - 1,
                        out[off:off + rows], beta if whole else 1)


def _transposed_for(matrix, out_lo: int, out_hi: int, src: torch.Tensor, src_lo: int) -> "TransposedBlock":
    t = getattr(matrix, "_transposed", None)
    if t is None or t.window != (out_lo, out_hi) or t.local_rows != src.numel() or matrix._transposed_origin != src_lo:
        if out_hi - out_lo + 1 > matrix.ctx.backend.COMPRESS_MAX_EXTENT or src.numel() > matrix.ctx.backend.COMPRESS_MAX_EXTENT:
            raise NotImplementedError("rmatvec: window or block of 2^32 - 1 elements or more")
        t = matrix._transposed = matrix.transpose(out_lo, out_hi, src_lo, src.numel())
        matrix._transposed_origin = src_lo
    return t


class _SlicedCopies:
    """What CSRMatrix and COOMatrix share around the sliced-pattern copies of their pieces
    (lsk_csr_sliced_*): making them for all pieces or none, choosing the placement of large ones by
    timing, re-making them a few times and keeping the fastest, and waiting until the driver is done
    scrubbing what that freed."""

    SLICED_REDRAWS = 6
    TUNE_MIN_ROWS = 1 << 21  # pieces below this are not worth the trial launches

    def _init_sliced(self) -> None:
        self.sliced = {}          # slice_structure(): piece colour -> sliced-pattern copy of the piece ...
        self.sliced_window = None # ... for products against this window
        self.sliced_placement = {}
        self.pair_placement = {}     # ... of the two-vector product, chosen at its first launch (matvec_pair)
        self._sliced_refill = {}  # piece colour -> refill(copy): the piece's current values into that copy
        self._sliced_make = None
        self.setup_seconds = {}   # wall-clock seconds of the set-up stages run on this matrix
        self.setup_notes = []     # what set-up had to leave out (a copy that did not fit in memory)
        register = getattr(self.ctx.backend, "register_releasable", None)
        if register is not None:
            register(self)        # (release_optional_memory: the copies go when a vector finds no room)

    def release_optional_memory(self) -> bool:
        """Called when an allocation on this device found no room: the set-up-time copies are what the matrix can
        work without.  The sliced-pattern copies go first (a copy of all values), the 32-bit copy of the
        structure is made in their place where that fits (half the bytes); asked again, that one goes as well
        and the products read the arrays as they were handed over.  Returns whether anything was freed."""
        if self.sliced:
            window, made_with = self.sliced_window, self._sliced_make
            for s in self.sliced.values():
                s.close()
            self.sliced, self.sliced_window, self._sliced_make = {}, None, None
            self.sliced_placement, self.pair_placement, self._sliced_refill = {}, {}, {}
            self.__dict__.pop("_placed_on_vectors", None)
            self.setup_notes.append("sliced-pattern copies given up: an allocation found no room")
            compress = getattr(self, "compress", None)
            if compress is not None and made_with is not None and getattr(self, "col32", None) is None:
                compress(window[0], window[1], made_with[1], made_with[2])
            return True
        if getattr(self, "col32", None) is not None and getattr(self, "rowoff", None):
            self.col32, self.col32_window, self.rowoff, self.long_rows = None, None, {}, {}
            torch.cuda.empty_cache()
            self.setup_notes.append("32-bit copy of the structure given up: an allocation found no room")
            return True
        return False

    def _refill_sliced_copies(self) -> None:
        """The sliced-pattern copies hold a copy of the entry values: bring them up to date."""
        for c, s in self.sliced.items():
            self._sliced_refill[c](s)

    def _stage(self, name: str):
        """Context manager: wall-clock seconds of a set-up stage (device work included), added to
        self.setup_seconds[name]."""
        matrix = self

        class _Timer:
            def __enter__(self):
                if matrix.val.is_cuda:
                    torch.cuda.synchronize(matrix.val.device)
                self.t0 = time.perf_counter()

            def __exit__(self, *exc):
                if matrix.val.is_cuda:
                    torch.cuda.synchronize(matrix.val.device)
                matrix.setup_seconds[name] = matrix.setup_seconds.get(name, 0.0) + time.perf_counter() - self.t0
                return False

        return _Timer()

    def _make_sliced_copies(self, make, x_lo: int, x_hi: int, kernel_partition, range_partition) -> bool:
        """make(i, c): the copy of local piece i (colour c), None when the library declines.  The
        copies are made for all pieces or none; unless LSK_TUNE=0, their placement is then chosen by
        timing (tune_sliced_copies; with LSK_TUNE=2 they are also re-made a few times)."""
        made = {}
        with self._stage("sliced_copy"):
            for i, c in enumerate(kernel_partition.colors):
                k_lo, k_hi = kernel_partition.intervals[i]
                if range_partition.pieces[c].volume == 0 or k_hi < k_lo:
                    continue
                try:
                    made[c] = make(i, c)
                except RuntimeError as exc:
                    if not _no_room(exc):
                        raise
                    made[c] = None   # the copy does not fit beside the matrix: the products go on without it
                    self.setup_notes.append(f"no room for the sliced-pattern copy of piece {c}: {exc}")
                if made[c] is None:
                    for s in made.values():
                        if s is not None:
                            s.close()
                    return False
        if not made:
            return False
        self.sliced, self.sliced_window = made, (int(x_lo), int(x_hi))
        self._sliced_make = (make, kernel_partition, range_partition)
        if os.environ.get("LSK_TUNE", "1") != "0":
            try:
                self.tune_sliced_copies()
            except RuntimeError as exc:   # (its two temporary vectors)
                if not _no_room(exc):
                    raise
                self.setup_notes.append(f"no room to time the placements: {exc}")
        return True

    def tune_sliced_copies(self, redraws: int = None) -> None:
        """Set-up-time speed tuning of the sliced-pattern copies of large pieces (TUNE_MIN_ROWS rows
        or more), none of which changes a result: (1) the workgroup -> XCD placement is chosen by
        timing a few candidates (0.1 s at the headline size); (2) OPT-IN, with LSK_TUNE=2 or a
        `redraws` argument: the copy is made up to `redraws` more times (LSK_TUNE_REDRAWS, default
        SLICED_REDRAWS) while the previous one is still held, and the fastest is kept: where the driver
        puts an array in HBM moves the product's time by several per cent, the draw cannot be steered
        but it can be repeated (needs room for a second copy, seconds of set-up; worth about 2 % at the
        headline size, for a solve of thousands of steps); (3) after redraws, a wait until the driver's
        scrubbing of what they freed no longer slows the product.  The planner runs it when it registers
        a matrix (unless LSK_TUNE=0); seconds per stage in self.setup_seconds."""
        if not self.sliced or not self.val.is_cuda or getattr(self, "_sliced_make", None) is None:
            return
        make, kernel_partition, range_partition = self._sliced_make
        x_lo, x_hi = self.sliced_window
        be = self.ctx.backend
        x = y = last = None
        freed_total = 0
        if redraws is None:
            redraws = int(os.environ.get("LSK_TUNE_REDRAWS", self.SLICED_REDRAWS)) if _redraws_asked_for() else 0
        for i, c in enumerate(kernel_partition.colors):
            rows = range_partition.pieces[c].volume
            if c not in self.sliced or rows < self.TUNE_MIN_ROWS:
                continue
            if x is None:
                x = torch.zeros(x_hi - x_lo + 1, dtype=self.dtype, device=self.val.device)
            if y is None or y.numel() < rows:
                y = torch.empty(rows, dtype=self.dtype, device=self.val.device)

            def best_of(s):
                return be.csr_sliced_tune(s, x, y)

            with self._stage("placement_tuner"):
                self.sliced_placement[c], best = best_of(self.sliced[c])
            freed = 0
            with self._stage("placement_redraws"):
                for _ in range(redraws):
                    try:
                        other = make(i, c)  # a second copy while the first is alive: another place in HBM
                    except RuntimeError:    # no room for it
                        break
                    if other is None:
                        break
                    pl, t = best_of(other)
                    freed += other.padded_entries * self.val.element_size()
                    if t < best:
                        self.sliced[c].close()
                        self.sliced[c], self.sliced_placement[c], best = other, pl, t
                        self.pair_placement.pop(c, None)
                    else:
                        other.close()
            freed_total += freed
            last = (c, rows, best)
        if freed_total >= (1 << 30):  # (smaller amounts are scrubbed before anyone could notice)
            with self._stage("steadiness_wait"):
                self._wait_until_steady(last[0], last[1], x, y, last[2], freed_total)

    SETTLE_SECONDS = 12.0

    def _wait_until_steady(self, c: int, rows: int, x: torch.Tensor, y: torch.Tensor, best: float,
                           freed: int) -> None:
        """The driver scrubs memory a process has freed in the background, at some 33 GB/s
        (measured: freeing 30 GB slows a running product by 2 % for 0.9 s; after the redraws above,
        which free ~29 GB each at the headline size, the product's time wanders between its best
        and +7 % for 5 s and then stays at the best for good).  So that whatever is timed next is
        not judged in that shadow, set-up ends by re-timing the product until it is back at the
        time it had when it was chosen, sixteen probes in a row (1.2 s; quiet spells of half a second
        occur inside the shadow), or twice the time the scrub should take (at most SETTLE_SECONDS) has
        passed."""
        limit = min(self.SETTLE_SECONDS, 1.5 + 2.0 * freed / 33e9)  # twice what the scrub should take
        deadline = time.monotonic() + limit
        good = 0
        while good < 16 and time.monotonic() < deadline:
            t = self.time_piece(0, c, x, y[:rows])
            good = good + 1 if t <= 1.01 * best else 0
            if good < 16:
                time.sleep(0.05)
        if os.environ.get("LSK_TUNE_VERBOSE"):
            print(f"[lsk] sliced product steady again {limit - (deadline - time.monotonic()):.1f} s "
                  f"after the redraws", file=sys.stderr, flush=True)

    def _place_on_these_vectors(self, c: int, sliced, rows: int, x: torch.Tensor, y: torch.Tensor) -> None:
        """The workgroup placement of a large piece's product, timed once more on the first vectors it is
        really multiplied with (every row of y is about to be overwritten anyway): which placement is fastest
        depends on where x and y lie in HBM as much as on the matrix -- on the 27-point 512^3 matrix the
        choice made on set-up's temporary vectors was 3 % off the best in one process of two
        (tools/exp/cg_by_placement.py) -- and a solver multiplies the same few vectors over and over (CG:
        always p -> q).  Once per piece; not while the stream records; LSK_TUNE=0 leaves it out."""
        done = self.__dict__.setdefault("_placed_on_vectors", set())
        if c in done or rows < self.TUNE_MIN_ROWS:
            return
        done.add(c)
        tuner = getattr(self.ctx.backend, "csr_sliced_tune", None)
        if (tuner is None or os.environ.get("LSK_TUNE", "1") == "0" or not x.is_cuda
                or torch.cuda.is_current_stream_capturing()):
            return
        with self._stage("placement_tuner"):
            self.sliced_placement[c], _ = tuner(sliced, x, y)

    def time_piece(self, i: int, c: int, x: torch.Tensor, y: torch.Tensor, reps: int = 3) -> float:
        """Milliseconds of the sliced-pattern product of local piece i (colour c) with the window x
        and the output piece y: `reps` launches after one untimed, and once more over enough launches for
        6 ms when the product is so short that those took less than 5 (so that 1 % shows above the jitter)."""
        be = self.ctx.backend
        rows = y.numel()
        t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        be.csr_sliced_spmv(self.sliced[c], 0, rows, x, y, 0, self.sliced_placement.get(c, 0))
        ms = 0.0
        for _ in range(2):
            t0.record()
            for _ in range(reps):
                be.csr_sliced_spmv(self.sliced[c], 0, rows, x, y, 0, self.sliced_placement.get(c, 0))
            t1.record()
            t1.synchronize()
            ms = t0.elapsed_time(t1) / reps
            if ms * reps >= 5.0 or reps >= 40:
                break
            reps = min(40, int(6.0 / max(ms, 0.15)) + 1)
        return ms


def _no_room(exc: BaseException) -> bool:
    """Whether a set-up step failed for lack of device memory (the library's hipMalloc or torch's allocator): the
    set-up-time copies are optional, a matrix that fills the GPU is multiplied on the arrays it came in."""
    return isinstance(exc, torch.cuda.OutOfMemoryError) or "OutOfMemory" in str(exc) or "out of memory" in str(exc)


def _redraws_asked_for() -> bool:
    """Re-making set-up-time copies to draw a better place in HBM is opt-in: LSK_TUNE=2."""
    return os.environ.get("LSK_TUNE", "1") == "2"


class CSRMatrix(AbstractMatrix, _SlicedCopies):
    def __init__(self, ctx, rects: torch.Tensor, k_base: int, col: torch.Tensor, val: torch.Tensor,
                 range_space: IndexSpace, row_lo: int):
        """rects: this rank's rows row_lo .. row_lo+rows-1; col/val: the kernel entries
        k_base .. k_base+nnz-1 those rects point into."""
        if rects.dtype != torch.int64 or col.dtype != torch.int64:
            raise TypeError("rects and col must be int64 (the reference's long long coordinates)")
        if rects.dim() != 2 or rects.shape[1] != 2 or not rects.is_contiguous():
            raise ValueError("rects must be a contiguous (rows, 2) tensor")
        if col.numel() != val.numel():
            raise ValueError("col and val differ in length")
        self.ctx = ctx
        self.rects, self.k_base, self.col, self.val = rects, int(k_base), col, val
        self._val_given = val  # set-up may move the arrays (tune): values_changed() reads this one
        self.dtype = val.dtype
        self.range_space = range_space
        self.row_lo = int(row_lo)
        self.rows = rects.shape[0]
        self.nnz = col.numel()
        self.placement = {}  # piece colour -> tile placement found by tune()
        self.col32 = None         # compress(): the column field as 32-bit offsets into ...
        self.col32_window = None  # ... this window (x_lo, x_hi)
        self.rowoff = {}          # ... and per piece colour rows + 1 32-bit offsets of the rows' entries
        self.long_rows = {}       # ... and, for pieces that have them, the plan for rows too long for one wave
        self._init_sliced()

    def values_changed(self) -> None:
        """To be called after the entries were rewritten in place, in the `val` tensor the matrix was
        made from (same structure, new values), before the next product.  The reference's task reads
        the region on every call (src/CSRMatrixTasks.cu:88-118); here set-up may have moved the arrays
        to another place in HBM (tune) and made copies of the values (slice_structure, the transpose
        behind rmatvec), all of which this brings up to date: one pass over the values per copy,
        stream-ordered."""
        if self.val is not self._val_given:
            self.ctx.backend.copy(self.val, self._val_given)
        self._refill_sliced_copies()
        self._transposed = None  # rebuilt by the next rmatvec

    def transpose(self, x_lo: int, x_hi: int, row_lo: int, rows: int) -> TransposedBlock:
        """The transpose of this rank's rows against the window x_lo..x_hi of the domain."""
        if row_lo != self.row_lo or rows != self.rows:
            raise ValueError("rmatvec: the source vector does not hold exactly this rank's rows")
        be = self.ctx.backend
        return TransposedBlock(be, self.col, be.csr_entry_rows(self.rects, self.k_base, self.nnz), self.val,
                               x_lo, x_hi, rows)

    def rmatvec(self, out, out_lo, out_hi, src, src_lo, beta):
        _transposed_for(self, out_lo, out_hi, src, src_lo).apply(self.ctx.backend, out, src, beta)

    FUSE_DOT_MAX_ROW_LENGTH = 16.0

    def fuses_dot(self) -> bool:
        """Whether the fused product + dot leaf pays on this matrix.  It costs 3-4 % of the
        product (measured: +0.4 ms on the 27-point 512^3 matrix, as much as the dot it saves)
        and saves a pass over two vectors, which is 10 % of a 5-/7-point product."""
        override = os.environ.get("LSK_FUSE_SPMV_DOT")
        if override is not None:
            return override != "0"
        if self.sliced:
            return True   # one lane owns a row there: the shares cost next to nothing
        if self.long_rows:
            return False  # the split product has no form that leaves dot shares
        return self.rows > 0 and self.nnz / self.rows <= self.FUSE_DOT_MAX_ROW_LENGTH

    def compress(self, x_lo: int, x_hi: int, kernel_partition, range_partition) -> bool:
        """Set-up-time copy of the matrix structure in 32 bits: the column field as offsets into
        the window x_lo..x_hi of the input vector (the ghost interval the planner found for this
        rank) and, per piece, the rects as rows + 1 offsets into the piece's entries.  Products
        against exactly that window then read 4 instead of 8 bytes of column and 4 instead of 16
        of row structure and give bit-identical results.  What the reference redoes on every
        call for cuSPARSE (src/CuSPARSEHelpers.hpp:9-31,71-80), done once.  LSK_COMPRESS=0 turns
        it off.  Returns whether the copy was made (not for windows of 2^32 - 1 columns or more,
        pieces of 2^32 entries or more, or rects that are not back to back)."""
        be = self.ctx.backend
        cols, rows_of = getattr(be, "csr_compress_cols", None), getattr(be, "csr_compress_rows", None)
        if (cols is None or rows_of is None or os.environ.get("LSK_COMPRESS", "1") == "0" or x_hi < x_lo
                or self.nnz == 0 or x_hi - x_lo + 1 > be.COMPRESS_MAX_EXTENT):
            return False
        try:
            return self._compress(be, cols, rows_of, x_lo, x_hi, kernel_partition, range_partition)
        except RuntimeError as exc:
            if not _no_room(exc):
                raise
            self.rowoff, self.long_rows, self.col32, self.col32_window = {}, {}, None, None
            self.setup_notes.append(f"no room for the 32-bit copy of the structure: {exc}")
            return False

    def _compress(self, be, cols, rows_of, x_lo, x_hi, kernel_partition, range_partition) -> bool:
        rowoff, long_rows = {}, {}
        with self._stage("compress_32bit"):
            for i, c in enumerate(kernel_partition.colors):
                p = range_partition.pieces[c]
                k_lo, k_hi = kernel_partition.intervals[i]
                nnz = max(k_hi - k_lo + 1, 0)
                if p.volume == 0:
                    continue
                r0 = p.lo[0] - self.row_lo
                off = rows_of(self.rects[r0:r0 + p.volume], k_lo if nnz else self.k_base, nnz)
                if off is None:
                    return False
                rowoff[c] = off
                planner = getattr(be, "csr_long_rows_create", None)
                if planner is not None and os.environ.get("LSK_SPLIT_ROWS", "1") != "0":
                    plan = planner(off)   # None for nearly every matrix: no row beyond a few thousand entries
                    if plan is not None:
                        long_rows[c] = plan
            self.rowoff, self.long_rows = rowoff, long_rows
            self.col32 = cols(self.col, x_lo, x_hi)
            self.col32_window = (int(x_lo), int(x_hi))
        return True

    def slice_structure(self, x_lo: int, x_hi: int, kernel_partition, range_partition) -> bool:
        """Set-up-time re-layout for matrices whose rows repeat a few column patterns (every
        linearised stencil): per piece a copy of the values in blocks of 512 (fp64) / 1024 (fp32) rows
        plus one pattern id per row; products against the window x_lo..x_hi then read no column array and no row
        structure, and sum every row in entry order with separate multiply and add -- bit for bit
        the loop of src/CSRMatrixTasks.cpp:73-91.  All pieces or none.  LSK_SLICED=0 turns it off.
        For large pieces the tile placement is then chosen by timing, and the copy is made up to
        SLICED_REDRAWS more times, keeping the fastest (where the driver puts an array in HBM
        moves the product's time by several per cent)."""
        be = self.ctx.backend
        maker = getattr(be, "csr_sliced_create", None)
        if maker is None or os.environ.get("LSK_SLICED", "1") == "0" or x_hi < x_lo or self.nnz == 0:
            return False

        def make(i, c):
            p = range_partition.pieces[c]
            k_lo, k_hi = kernel_partition.intervals[i]
            nnz = max(k_hi - k_lo + 1, 0)
            r0, off = p.lo[0] - self.row_lo, k_lo - self.k_base
            self._sliced_refill[c] = lambda s, rows=p.volume: be.csr_sliced_update_values(
                s, rows, nnz, self.rects[r0:r0 + rows], k_lo, self.val[off:off + nnz])
            return maker(p.volume, nnz, self.rects[r0:r0 + p.volume], k_lo, self.col[off:off + nnz],
                         self.val[off:off + nnz], p.lo[0], x_lo, x_hi)
        return self._make_sliced_copies(make, x_lo, x_hi, kernel_partition, range_partition)

    def _structure(self, x_lo: int, x_hi: int, c: int, r0: int, ra: int, rb: int):
        """(row field, column array) a product of rows [ra, rb) of piece c (whose first row is
        this rank's row r0) against the window x_lo..x_hi reads."""
        if self.col32 is not None and self.col32_window == (x_lo, x_hi) and c in self.rowoff:
            return self.rowoff[c][ra:rb + 1], self.col32
        return self.rects[r0 + ra:r0 + rb], self.col

    def tune(self, x_lo, x_hi, kernel_partition, range_partition) -> None:
        """Set-up-time choice of the tile -> XCD placement per piece (the backend times a few
        candidates on the piece itself).  A speed hint: results do not depend on it.  No
        counterpart in the reference, where cuSPARSE picks its algorithm internally."""
        tuner = getattr(self.ctx.backend, "csr_spmv_tune", None)
        if tuner is None or os.environ.get("LSK_TUNE", "1") == "0" or x_hi < x_lo:
            return
        x = y = None
        big = [i for i, c in enumerate(kernel_partition.colors)
               if range_partition.pieces[c].volume >= self.TUNE_MIN_ROWS
               and kernel_partition.intervals[i][1] >= kernel_partition.intervals[i][0]]
        if not big:
            return
        try:
            x = torch.zeros(x_hi - x_lo + 1, dtype=self.dtype, device=self.val.device)
            y = torch.empty(max(range_partition.pieces[kernel_partition.colors[i]].volume for i in big),
                            dtype=self.dtype, device=self.val.device)
        except RuntimeError as exc:
            if not _no_room(exc):
                raise
            self.setup_notes.append(f"no room to time the placements: {exc}")
            return
        with self._stage("placement_redraws"):
            self._redraw_arrays(x, x_lo, x_hi, y, kernel_partition, range_partition, big)
        self._tune_placement(tuner, x, x_lo, x_hi, y, kernel_partition, range_partition, big)

    def _tune_placement(self, tuner, x, x_lo, x_hi, y, kernel_partition, range_partition, big) -> None:
        for i, c in enumerate(kernel_partition.colors):
            p = range_partition.pieces[c]
            k_lo, k_hi = kernel_partition.intervals[i]
            nnz = max(k_hi - k_lo + 1, 0)
            if i not in big:
                continue
            r0 = p.lo[0] - self.row_lo
            off = k_lo - self.k_base
            rowf, col = self._structure(x_lo, x_hi, c, r0, 0, p.volume)
            self.placement[c], _ = tuner(p.volume, nnz, rowf, k_lo,
                                         col[off:off + nnz], self.val[off:off + nnz], x, x_lo, x_hi, y)

    def halo_split(self, own_lo: int, own_hi: int, kernel_partition, range_partition):
        """For every local piece, its rows in three runs: [0, a) read ghost cells below own_lo,
        [a, rows - b) read only own_lo..own_hi, [rows - b, rows) read ghost cells above own_hi;
        each run with the kernel interval its rects cover.  The middle runs can be multiplied
        while the halo is still in flight.  Returns {piece: [(phase, r_begin, r_end, k_lo, k_hi)]}
        with phase 0 = needs no ghost cell, 1 = does, or None when the rects of a piece are not
        back to back (then a run's kernel interval is not simply its first and last rect)."""
        be = self.ctx.backend
        finder = getattr(be, "csr_halo_rows", None)
        if finder is None:
            return None
        out = {}
        for i, c in enumerate(kernel_partition.colors):
            p = range_partition.pieces[c]
            rows = p.volume
            k_lo, k_hi = kernel_partition.intervals[i]
            nnz = max(k_hi - k_lo + 1, 0)
            if rows == 0 or nnz == 0:
                continue
            r0 = p.lo[0] - self.row_lo
            rects = self.rects[r0:r0 + rows]
            if rows > 1 and not bool((rects[1:, 0] == rects[:-1, 1] + 1).all()):
                return None
            off = k_lo - self.k_base
            below, above = finder(rows, nnz, rects, k_lo, self.col[off:off + nnz], own_lo, own_hi)
            a, b0 = below + 1, above          # [0, a) and [b0, rows) need ghost cells
            if a >= b0:                       # they meet: nothing to run early
                out[i] = [(1, 0, rows, k_lo, k_hi)]
                continue
            cuts = [0, a, b0, rows]
            edge = rects[[min(a, rows - 1), min(b0, rows - 1)], 0].tolist()  # first entries of rows a, b0
            k_cut = [k_lo, edge[0] if a < rows else k_hi + 1, edge[1] if b0 < rows else k_hi + 1, k_hi + 1]
            runs = []
            for j, phase in enumerate((1, 0, 1)):
                if cuts[j + 1] > cuts[j]:
                    runs.append((phase, cuts[j], cuts[j + 1], k_cut[j], k_cut[j + 1] - 1))
            out[i] = runs
        return out

    REDRAWS = 6  # extra placements of the arrays tried at set-up (draws repeat after a handful)

    def _redraw_arrays(self, x, x_lo, x_hi, y, kernel_partition, range_partition, pieces) -> None:
        """Where the driver puts rects/col/val in HBM moves the product's time by +-6 % (same
        kernel, same virtual addresses, re-allocated arrays: 10.3 .. 11.7 ms on the 27-point 512^3
        matrix; x and y do not matter).  The draw cannot be steered, but it can be repeated: copy
        the arrays into fresh allocations while the old ones are still held, time both, keep the
        faster set, a few times.  Needs room for a second copy; skipped otherwise."""
        redraws = int(os.environ.get("LSK_TUNE_REDRAWS", str(self.REDRAWS))) if _redraws_asked_for() else 0
        if redraws <= 0 or not self.val.is_cuda or getattr(self.ctx.backend, "csr_spmv_tune", None) is None:
            return
        compressed = self.col32 is not None and self.col32_window == (x_lo, x_hi)  # then rects / col are off the path
        need = sum(t.numel() * t.element_size()
                   for t in ((self.col32, self.val, *self.rowoff.values()) if compressed
                             else (self.rects, self.col, self.val)))

        def time_current() -> float:
            def sweep():
                for i in pieces:
                    c = kernel_partition.colors[i]
                    p = range_partition.pieces[c]
                    k_lo, k_hi = kernel_partition.intervals[i]
                    r0, off, nnz = p.lo[0] - self.row_lo, k_lo - self.k_base, k_hi - k_lo + 1
                    rowf, col = self._structure(x_lo, x_hi, c, r0, 0, p.volume)
                    self.ctx.backend.csr_spmv(p.volume, nnz, rowf, k_lo,
                                              col[off:off + nnz], self.val[off:off + nnz], x, x_lo, x_hi,
                                              y, 0, self.placement.get(c, 0))
            for _ in range(2):
                sweep()
            t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            t0.record()
            for _ in range(5):
                sweep()
            t1.record()
            t1.synchronize()
            return t0.elapsed_time(t1) / 5

        best = time_current()
        for _ in range(redraws):
            free, _total = torch.cuda.mem_get_info(self.val.device)
            if free < need + (2 << 30):
                break
            if compressed:
                old = (self.rowoff, self.col32, self.val)
                self.rowoff = {c: t.clone() for c, t in self.rowoff.items()}
                self.col32, self.val = self.col32.clone(), self.val.clone()
            else:
                old = (self.rects, self.col, self.val)
                self.rects, self.col, self.val = (t.clone() for t in old)
            t = time_current()
            if os.environ.get("LSK_TUNE_VERBOSE"):
                print(f"[lsk] array placement: {best:.3f} ms held, {t:.3f} ms redrawn", file=sys.stderr, flush=True)
                if os.environ.get("LSK_TUNE_VERBOSE") == "2":
                    col_now = self.col32 if compressed else self.col
                    print(f"[lsk]   redrawn at val {self.val.data_ptr():#x} col {col_now.data_ptr():#x} "
                          f"(val - col = {(self.val.data_ptr() - col_now.data_ptr()) / 2**30:.3f} GiB)",
                          file=sys.stderr, flush=True)
            if t < best:
                best = t
            elif compressed:
                self.rowoff, self.col32, self.val = old
            else:
                self.rects, self.col, self.val = old
            del old
            torch.cuda.empty_cache()  # hand the losing set back to the driver before the next draw
        # the driver scrubs what was just freed in the background, which slows whatever runs
        # meanwhile (see _wait_until_steady): leave once the product is back at its time
        deadline, good = time.monotonic() + self.SETTLE_SECONDS, 0
        while need >= (1 << 30) and good < 4 and time.monotonic() < deadline:
            good = good + 1 if time_current() <= 1.01 * best else 0

    def create_kernel_partition_from_range_partition(self, range_partition):
        """Image-range of the rowptr rects over each piece's rows (src/CSRMatrix.cpp:89-109)."""
        colors = range_partition.colors_of_rank(self.ctx.rank, self.ctx.world_size)
        out = []
        for c in colors:
            p = range_partition.pieces[c]
            lo, hi = p.lo[0] - self.row_lo, p.hi[0] - self.row_lo
            if hi < lo:
                out.append((0, -1))
                continue
            if lo < 0 or hi >= self.rows:
                raise ValueError("range piece lies outside this rank's rows")
            out.append(self.ctx.backend.rect_bounds(self.rects[lo:hi + 1]))
        return PiecePartition(colors, out)

    def create_domain_partition_from_kernel_partition(self, domain_space, kernel_partition):
        """Image of the column field over each kernel piece (src/CSRMatrix.cpp:112-132)."""
        return _ghost_from_kernel(self.ctx, self.col, self.k_base, domain_space, kernel_partition)

    def matvec(self, dst, x, x_lo, x_hi, kernel_partition, beta, range_partition=None, dot_with=None,
               split=None, phase=None):
        """CSRMatrix::matvec (src/CSRMatrix.cpp:158-214): one csr_matvec leaf per piece.

        dot_with: a vector on dst's partition; the fused product + dot leaf then also leaves
        the partial sums of dot(dst, dot_with) with dst (a speed hint for the dot product a
        Krylov step takes next; only with beta = 0).
        split, phase: with the result of halo_split, launch only the runs of rows of that phase
        (0: those that need no ghost cell, 1: the others); both phases together are the whole
        product (and, on the sliced-pattern copy, leave all the shares of the dot)."""
        range_partition = range_partition or dst.index_partition
        be = self.ctx.backend
        sliced = self.sliced if self.sliced_window == (x_lo, x_hi) else {}
        fused = (dot_with is not None and beta == 0 and (split is None or bool(sliced))
                 and getattr(be, "csr_spmv_dot", None) is not None)
        last_phase = split is None or phase == 1
        touch = getattr(dst, "_touch", None)
        if touch is not None:
            touch()
        pieces = []
        for i, c in enumerate(kernel_partition.colors):
            p = range_partition.pieces[c]
            rows = p.volume
            if rows == 0:
                pieces.append((None, 0))
                continue
            r0 = p.lo[0] - self.row_lo
            k_lo, k_hi = kernel_partition.intervals[i]
            nnz = max(k_hi - k_lo + 1, 0)
            if c in sliced and beta == 0:
                self._place_on_these_vectors(c, sliced[c], rows, x, dst._piece(i))
            if split is not None and i in split:
                if fused:  # every run's shares side by side in one buffer, whichever phase launches it
                    slots = ([be.csr_sliced_dot_slots(sliced[c], rb, ra) for _, ra, rb, _, _ in split[i]] if c in sliced
                             else [0] * len(split[i]))
                    buf = dst._partials("mv", i, sum(slots))
                    if last_phase:
                        pieces.append((buf, sum(slots)))
                for k, (run_phase, ra, rb, ka, kb) in enumerate(split[i]):
                    if run_phase != phase:
                        continue
                    n_run = max(kb - ka + 1, 0)
                    if c in sliced and fused:
                        be.csr_sliced_spmv_dot(sliced[c], rb, x, dst._piece(i), beta, dot_with._piece(i),
                                               buf[sum(slots[:k]):], self.sliced_placement.get(c, 0), row_begin=ra)
                        continue
                    if c in sliced:
                        be.csr_sliced_spmv(sliced[c], ra, rb, x, dst._piece(i), beta, self.sliced_placement.get(c, 0))
                        continue
                    rowf, col = self._structure(x_lo, x_hi, c, r0, ra, rb)
                    if rowf.dtype == torch.int32:   # offsets count from the piece's first entry
                        off, n_arr = (k_lo - self.k_base) if nnz else 0, nnz
                    else:
                        off, n_arr = (ka - self.k_base) if n_run else 0, n_run
                    # (nnz = the entries of THESE rows: the kernel sizes its tiles by nnz / rows)
                    if rowf.dtype == torch.int32 and c in self.long_rows:
                        be.csr_spmv_split(self.long_rows[c], ra, rb - ra, n_run, rowf, col[off:off + n_arr],
                                          self.val[off:off + n_arr], x, x_lo, x_hi, dst._piece(i)[ra:rb], beta,
                                          self.placement.get(c, 0))
                        continue
                    be.csr_spmv(rb - ra, n_run, rowf, ka if n_run else self.k_base,
                                col[off:off + n_arr], self.val[off:off + n_arr], x, x_lo, x_hi,
                                dst._piece(i)[ra:rb], beta, self.placement.get(c, 0))
                continue
            if split is not None and phase == 0:
                continue  # a piece without a split entry (no entries) goes with the late phase
            if c in sliced and fused:
                buf = dst._partials("mv", i, be.csr_sliced_dot_slots(sliced[c], rows))
                pieces.append((buf, be.csr_sliced_spmv_dot(sliced[c], rows, x, dst._piece(i), beta, dot_with._piece(i),
                                                           buf, self.sliced_placement.get(c, 0))))
                continue
            if c in sliced:
                be.csr_sliced_spmv(sliced[c], 0, rows, x, dst._piece(i), beta, self.sliced_placement.get(c, 0))
                continue
            off = (k_lo - self.k_base) if nnz else 0
            rowf, col = self._structure(x_lo, x_hi, c, r0, 0, rows)
            args = (rows, nnz, rowf, k_lo if nnz else self.k_base,
                    col[off:off + nnz], self.val[off:off + nnz], x, x_lo, x_hi, dst._piece(i), beta)
            if fused:
                buf = dst._partials("mv", i, be.csr_spmv_dot_slots(rows, nnz, self.dtype))
                pieces.append((buf, be.csr_spmv_dot(*args, dot_with._piece(i), buf, self.placement.get(c, 0))))
            elif rowf.dtype == torch.int32 and c in self.long_rows:
                be.csr_spmv_split(self.long_rows[c], 0, rows, nnz, rowf, col[off:off + nnz], self.val[off:off + nnz],
                                  x, x_lo, x_hi, dst._piece(i), beta, self.placement.get(c, 0))
            else:
                be.csr_spmv(*args, self.placement.get(c, 0))
        if fused and last_phase:
            dst._note_ready(dot_with, pieces)


    def pairs(self, x_lo, x_hi, kernel_partition, range_partition) -> bool:
        """Whether matvec_pair serves this matrix for the window: every local piece that has rows is
        multiplied through its sliced-pattern copy."""
        be = self.ctx.backend
        return (getattr(be, "csr_sliced_spmv2", None) is not None and os.environ.get("LSK_PAIR", "1") != "0"
                and self.sliced_window == (x_lo, x_hi)
                and all(c in self.sliced and self.sliced[c].pair_served
                        for c in kernel_partition.colors if range_partition.pieces[c].volume > 0))

    def matvec_pair(self, dsts, xs, x_lo, x_hi, kernel_partition, dot_with=None, range_partition=None,
                    split=None, phase=None) -> bool:
        """dsts[k] <- A xs[k] for k = 0, 1 with ONE pass over the values (two systems on one matrix: what
        the reference's BenchmarkStencil steps, test/BenchmarkStencil.cpp:181-199, where the planner
        launches one CSRMatvecTask per system).  Each result is bit for bit that of matvec(..., beta = 0);
        dot_with = (w0, w1) leaves the shares of dot(dsts[k], w_k) as matvec's dot_with does; split, phase
        as in matvec (the runs of rows of one phase; both phases together are the whole product and
        leave all the shares).  False when a piece's structure is not served: nothing was launched for
        ANY piece, the caller multiplies one by one."""
        range_partition = range_partition or dsts[0].index_partition
        be = self.ctx.backend
        if not self.pairs(x_lo, x_hi, kernel_partition, range_partition):
            return False
        fused = dot_with is not None
        last_phase = split is None or phase == 1
        launched = False
        pieces = ([], [])
        for i, c in enumerate(kernel_partition.colors):
            rows = range_partition.pieces[c].volume
            if rows == 0:
                for k in (0, 1):
                    pieces[k].append((None, 0))
                continue
            sliced = self.sliced[c]
            ys = [d._piece(i) for d in dsts]
            ws = [w._piece(i) for w in dot_with] if fused else None
            recording = ys[0].is_cuda and torch.cuda.is_current_stream_capturing()
            if c not in self.pair_placement and recording:
                place = self.sliced_placement.get(c, 0) or 64   # timing waits for the device: after the recording
            elif c not in self.pair_placement:
                # Two windows of x compete for each XCD's L2, so the pair has a placement of its own, timed
                # once on the vectors at hand (every row of ys is about to be overwritten anyway).
                # (untimed: runs of 64 blocks per XCD -- only the round-robin placement is slow for the pair)
                self.pair_placement[c] = self.sliced_placement.get(c, 0) or 64
                tune2 = getattr(be, "csr_sliced_tune2", None)
                if tune2 is not None and rows >= self.TUNE_MIN_ROWS and os.environ.get("LSK_TUNE", "1") != "0":
                    with self._stage("pair_placement_tuner"):
                        tuned = tune2(sliced, xs, ys)
                    if tuned is not None:
                        self.pair_placement[c] = tuned[0]
            if c in self.pair_placement:
                place = self.pair_placement[c]
            if split is not None and i in split:
                runs = split[i]
                slots = [be.csr_sliced_dot_slots(sliced, rb, ra) for _, ra, rb, _, _ in runs] if fused else []
                # every run's shares side by side in one buffer per system, whichever phase launches it
                bufs = [d._partials("mv", i, sum(slots)) for d in dsts] if fused else None
                for n, (run_phase, ra, rb, _, _) in enumerate(runs):
                    if run_phase != phase or rb <= ra:
                        continue
                    part = [b[sum(slots[:n]):] for b in bufs] if fused else None
                    if be.csr_sliced_spmv2(sliced, rb, xs, ys, 0, ws, part, place, row_begin=ra) is None:
                        raise RuntimeError("the two-vector product declined rows of a structure it serves")
                launched = True
                for k in (0, 1):
                    pieces[k].append((bufs[k] if fused else None, sum(slots)))
                continue
            if split is not None and phase == 0:
                for k in (0, 1):   # (a piece without a split entry goes with the late phase)
                    pieces[k].append((None, 0))
                continue
            bufs = None
            if fused:
                slots = be.csr_sliced_dot_slots(sliced, rows)
                bufs = [d._partials("mv", i, slots) for d in dsts]
            count = be.csr_sliced_spmv2(sliced, rows, xs, ys, 0, ws, bufs, place)
            if count is None:
                if launched:
                    raise RuntimeError("the two-vector product served one piece and declined another")
                return False
            launched = True
            for k in (0, 1):
                pieces[k].append((bufs[k] if fused else None, count))
        for d in dsts:
            touch = getattr(d, "_touch", None)
            if touch is not None:
                touch()
        if fused and last_phase:
            for k in (0, 1):
                dsts[k]._note_ready(dot_with[k], pieces[k])
        return True


class COOMatrix(AbstractMatrix, _SlicedCopies):
    def __init__(self, ctx, row: torch.Tensor, col: torch.Tensor, val: torch.Tensor,
                 range_space: IndexSpace):
        """This rank's entries: every entry whose row this rank owns (the preimage of its
        range pieces).  Row-sorted entries give one contiguous kernel piece per range piece."""
        if row.dtype != torch.int64 or col.dtype != torch.int64:
            raise TypeError("row and col must be int64")
        if not (row.numel() == col.numel() == val.numel()):
            raise ValueError("row, col and val differ in length")
        self.ctx = ctx
        self.row, self.col, self.val = row, col, val
        self.dtype = val.dtype
        self.range_space = range_space
        self.nnz = val.numel()
        self.row_sorted = bool((row[1:] >= row[:-1]).all()) if self.nnz > 1 else True
        self.blocked = False            # entries of the 32-bit copies re-ordered by column blocks
        self.val_blocked = None         # ... with the values in that order
        self._block_order = None        # ... which is val[_block_order]
        self.row32 = self.col32 = None  # compress(): 32-bit copies of the index fields ...
        self.col32_window = None        # ... for this x window (x_lo, x_hi) ...
        self.row_origin = 0             # ... and rows counted from this global row
        self.binned = None              # the binned copy (lsk_coo_binned_*) of the one local piece ...
        self.binned_window = None       # ... for this x window
        self.binned_color = None
        self._init_sliced()

    def release_optional_memory(self) -> bool:
        """As CSRMatrix's: the sliced-pattern copies first, then the binned copy (the products then scan the
        entries in the order they came in); the 32-bit index fields stay."""
        if _SlicedCopies.release_optional_memory(self):
            return True
        if self.binned is not None:
            self.binned.close()
            self.binned = self.binned_window = self.binned_color = None
            self.setup_notes.append("binned copy given up: an allocation found no room")
            return True
        return False

    def compress(self, x_lo: int, x_hi: int, kernel_partition, range_partition) -> bool:
        """Set-up-time 32-bit copies of the index fields: columns as offsets into the window
        x_lo..x_hi of the input vector, rows as offsets from this rank's first row (0xffffffff
        for an id outside its window, which every product drops as it drops the 64-bit id).
        16 instead of 24 bytes per fp64 entry; same kernel, same run sums.  LSK_COMPRESS=0 turns
        it off; not made for windows of 2^32 - 1 ids or more."""
        be = self.ctx.backend
        maker = getattr(be, "csr_compress_cols", None)
        owned = [range_partition.pieces[c] for c in kernel_partition.colors if range_partition.pieces[c].volume > 0]
        if (maker is None or os.environ.get("LSK_COMPRESS", "1") == "0" or x_hi < x_lo or self.nnz == 0
                or not owned):
            return False
        row_lo, row_hi = min(p.lo[0] for p in owned), max(p.hi[0] for p in owned)
        if max(x_hi - x_lo + 1, row_hi - row_lo + 1) > be.COMPRESS_MAX_EXTENT:
            return False
        self.row32 = maker(self.row, row_lo, row_hi)
        self.col32 = maker(self.col, x_lo, x_hi)
        self.col32_window = (int(x_lo), int(x_hi))
        self.row_origin = int(row_lo)
        self._try_column_blocks(x_lo, x_hi, kernel_partition, range_partition)
        return True

    def transpose(self, x_lo: int, x_hi: int, row_lo: int, rows: int) -> TransposedBlock:
        """The transpose of the entries whose row lies in row_lo..row_lo+rows-1 (this rank's rows)
        against the window x_lo..x_hi of the domain."""
        be = self.ctx.backend
        return TransposedBlock(be, self.col, be.csr_compress_cols(self.row, row_lo, row_lo + rows - 1), self.val,
                               x_lo, x_hi, rows)

    def rmatvec(self, out, out_lo, out_hi, src, src_lo, beta):
        _transposed_for(self, out_lo, out_hi, src, src_lo).apply(self.ctx.backend, out, src, beta)

    def slice_structure(self, x_lo: int, x_hi: int, kernel_partition, range_partition) -> bool:
        """The sliced-pattern copy (see CSRMatrix.slice_structure) for row-sorted COO matrices whose
        rows repeat a few column patterns: the rows' runs of entries are what a CSR matrix calls its
        rects.  One lane then sums a row's entries in entry order onto what y holds, which for
        row-sorted entries is bit for bit the loop of src/COOMatrixTasks.cpp:67-74.  A matrix with
        a column outside the window, unsorted rows or no such structure keeps the forms above."""
        be = self.ctx.backend
        maker = getattr(be, "csr_sliced_create", None)
        if (maker is None or os.environ.get("LSK_SLICED", "1") == "0" or x_hi < x_lo or self.nnz == 0
                or not self.row_sorted):
            return False

        def make(i, c):
            p = range_partition.pieces[c]
            k_lo, k_hi = kernel_partition.intervals[i]
            nnz = k_hi - k_lo + 1
            def row_rects():
                bounds = torch.arange(p.lo[0], p.hi[0] + 2, dtype=torch.int64, device=self.row.device)
                first = torch.searchsorted(self.row[k_lo:k_lo + nnz], bounds) + k_lo   # rows + 1 entry offsets
                return torch.stack((first[:-1], first[1:] - 1), dim=1).contiguous()

            self._sliced_refill[c] = lambda s: be.csr_sliced_update_values(
                s, p.volume, nnz, row_rects(), k_lo, self.val[k_lo:k_lo + nnz])
            return maker(p.volume, nnz, row_rects(), k_lo, self.col[k_lo:k_lo + nnz], self.val[k_lo:k_lo + nnz],
                         p.lo[0], x_lo, x_hi)

        return self._make_sliced_copies(make, x_lo, x_hi, kernel_partition, range_partition)

    BLOCK_BYTES = 4 << 20        # of x per column block: what one XCD's L2 holds beside the stream
    BLOCK_MIN_WINDOW = 16 << 20  # smaller x windows are not worth re-ordering for
    BINNED_BLOCK_COLS = 0        # columns per block / rows per tile of the binned copy: 0 = the library's choice
    BINNED_TILE_ROWS = 0

    def _try_column_blocks(self, x_lo, x_hi, kernel_partition, range_partition) -> None:
        """For a matrix whose x gathers have no locality (sampled: nearly every entry of a run
        touches a cache line of its own) and whose x window is far larger than L2, time the
        product in two more forms and keep the fastest: the binned copy (two streaming passes with
        a block of x, then a tile of y, in LDS; lsk_coo_binned_*) and the scan kernel on entries
        re-ordered by blocks of columns (a block of x then stays in L2 while its entries stream
        by).  Measured on the random 32-per-row matrix of 15.6 M rows: 9.7 ms in row order, 5.1 ms
        column-blocked, 2.6 ms binned.  Only with one local piece (a re-ordered matrix has no
        contiguous run of entries per piece).  LSK_COO_BLOCK=0 turns both off, =1 skips the probe
        and the timing of the row order; LSK_COO_BINNED=0 leaves the binned copy out, =2 takes it
        without timing anything."""
        be = self.ctx.backend
        mode = os.environ.get("LSK_COO_BLOCK", "auto")
        esz = self.val.element_size()
        if (mode == "0" or getattr(be, "coo_block_order", None) is None or len(kernel_partition.colors) != 1
                or not self.val.is_cuda or self.nnz < (1 << 16)
                or (x_hi - x_lo + 1) * esz < self.BLOCK_MIN_WINDOW):
            return
        if mode != "1":
            mid = self.nnz // 2
            sample = self.col32[mid:mid + (1 << 16)].to(torch.int64) & 0xFFFFFFFF
            lines = torch.unique(sample // (128 // esz)).numel()
            if lines < 0.5 * sample.numel():
                return  # neighbouring entries share cache lines of x: the row order is fine
        c = kernel_partition.colors[0]
        p = range_partition.pieces[c]
        y_lo, y_hi = p.lo[0] - self.row_origin, p.hi[0] - self.row_origin
        x = torch.zeros(x_hi - x_lo + 1, dtype=self.dtype, device=self.val.device)
        y = torch.zeros(p.volume, dtype=self.dtype, device=self.val.device)

        def timed(launch) -> float:
            t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            launch()
            t0.record()
            for _ in range(3):
                launch()
            t1.record()
            t1.synchronize()
            return t0.elapsed_time(t1) / 3

        def scan(row32, col32, val):
            return lambda: be.coo_spmv(self.nnz, row32, col32, val, x, x_lo, x_hi, y, y_lo, y_hi)

        times = {}
        if mode != "1":
            times["rows"] = timed(scan(self.row32, self.col32, self.val))
        binned = None
        binned_mode = os.environ.get("LSK_COO_BINNED", "1")
        if binned_mode != "0" and getattr(be, "coo_binned_create", None) is not None:
            # (an optional speed probe: when the device has no room for the copy and its sort scratch,
            # 16 + 24 bytes per entry, set-up goes on with the other forms)
            try:
                binned = be.coo_binned_create(self.row32, self.col32, self.val, y_lo, y_hi, x_hi - x_lo + 1,
                                              self.BINNED_BLOCK_COLS, self.BINNED_TILE_ROWS)
                if binned is not None:
                    times["binned"] = 0.0 if binned_mode == "2" else timed(lambda: be.coo_binned_spmv(binned, x, y, 0))
            except RuntimeError:
                if binned is not None:
                    binned.close()
                binned = None
                times.pop("binned", None)
                torch.cuda.empty_cache()
        if binned is not None and binned_mode != "2" and times.get("rows", float("inf")) < times["binned"]:
            binned.close()   # it has lost already: lower the peak before the blocked copies are made
            binned = None
            del times["binned"]
        order = None
        blocked = None
        if not (binned is not None and binned_mode == "2"):
            try:
                order = be.coo_block_order(self.col32, x_hi - x_lo + 1, self.BLOCK_BYTES // esz)
                if order is not None:
                    blocked = tuple(be.gather(t, order) for t in (self.row32, self.col32, self.val))
                    times["column blocks"] = timed(scan(*blocked))
            except RuntimeError:
                order = blocked = None
                times.pop("column blocks", None)
                torch.cuda.empty_cache()
        if os.environ.get("LSK_TUNE_VERBOSE"):
            print("[lsk] COO forms: " + ", ".join(f"{k} {v:.3f} ms" for k, v in times.items()),
                  file=sys.stderr, flush=True)
        best = min(times, key=times.get) if times else "rows"
        if best == "binned":
            self.binned = binned
            self.binned_window = (int(x_lo), int(x_hi))
            self.binned_color = c
        elif best == "column blocks":
            self.row32, self.col32, self.val_blocked = blocked
            self.blocked = True
            self._block_order = order  # kept for values_changed(): 4 bytes per entry
        if best != "binned" and binned is not None:
            binned.close()

    def values_changed(self) -> None:
        """To be called after the entries were rewritten in place in the `val` tensor (same rows and
        columns, new values), before the next product: brings the set-up-time copies of the values up
        to date (the column-blocked order, the sliced-pattern copies, the transpose behind rmatvec).
        The reference's task reads the region on every call (src/COOMatrixTasks.cu:86-145)."""
        if self.blocked:
            self.val_blocked = self.ctx.backend.gather(self.val, self._block_order)
        if self.binned is not None:
            self.ctx.backend.coo_binned_update_values(self.binned, self.val)
        self._refill_sliced_copies()
        self._transposed = None

    def create_kernel_partition_from_range_partition(self, range_partition):
        """Preimage of the row field (src/COOMatrix.cpp:77-95).  For row-sorted entries each
        piece is the contiguous run of its rows; otherwise every piece spans all local
        entries and the leaf's row guard selects (src/COOMatrixTasks.cpp:71)."""
        colors = range_partition.colors_of_rank(self.ctx.rank, self.ctx.world_size)
        out = []
        for c in colors:
            p = range_partition.pieces[c]
            if p.volume == 0 or self.nnz == 0:
                out.append((0, -1))
            elif self.row_sorted:
                bounds = torch.tensor([p.lo[0], p.hi[0] + 1], dtype=torch.int64, device=self.row.device)
                lo, hi = torch.searchsorted(self.row, bounds).tolist()
                out.append((int(lo), int(hi) - 1))
            else:
                out.append((0, self.nnz - 1))
        return PiecePartition(colors, out)

    def create_domain_partition_from_kernel_partition(self, domain_space, kernel_partition):
        return _ghost_from_kernel(self.ctx, self.col, 0, domain_space, kernel_partition)

    def overwrites(self, x_lo, x_hi, kernel_partition, range_partition) -> bool:
        """Whether matvec(beta = 0) is available: every local piece that has rows is multiplied
        through its sliced-pattern copy, which can start a row's sum from zero (the zero fill the
        reference does before an accumulating product, src/SquarePlanner.hpp:341, then falls away)."""
        live = [c for c in kernel_partition.colors if range_partition.pieces[c].volume > 0]
        if self.binned is not None and self.binned_window == (x_lo, x_hi) and live == [self.binned_color]:
            return True  # (the binned copy's second pass writes every row of the piece)
        return self.sliced_window == (x_lo, x_hi) and all(c in self.sliced for c in live)

    def phases(self, x_lo, x_hi, kernel_partition, range_partition) -> bool:
        """Whether matvec(phase=...) is available for this window: the one local piece goes through its
        binned copy, whose first pass can start on the columns that are already there."""
        live = [c for c in kernel_partition.colors if range_partition.pieces[c].volume > 0]
        return (self.binned is not None and self.binned_window == (x_lo, x_hi) and live == [self.binned_color]
                and getattr(self.ctx.backend, "coo_binned_spmv_phase", None) is not None
                and os.environ.get("LSK_OVERLAP", "1") != "0")

    def matvec(self, dst, x, x_lo, x_hi, kernel_partition, beta=1, range_partition=None, phase=None, own=None):
        """COOMatrix::matvec (src/COOMatrix.cpp:144-191): accumulates; dst must hold the
        values to add to (the planner zero-fills it).  beta = 0 where overwrites() says so.
        phase, own (only where phases() says so): phase 0 multiplies what reads only the columns
        own = (lo, hi) (global; the ones this rank holds before the ghost cells have arrived), phase 1
        the rest and writes dst; both in that order are the whole product."""
        range_partition = range_partition or dst.index_partition
        if phase is not None:
            if not self.phases(x_lo, x_hi, kernel_partition, range_partition) or own is None:
                raise ValueError("this COO product cannot be taken in phases")
            i = kernel_partition.colors.index(self.binned_color)
            if phase == 1 and getattr(dst, "_touch", None) is not None:
                dst._touch()
            self.ctx.backend.coo_binned_spmv_phase(self.binned, x, dst._piece(i), beta, phase, own[0] - x_lo,
                                                   own[1] - x_lo)
            return
        if beta != 1 and not (beta == 0 and self.overwrites(x_lo, x_hi, kernel_partition, range_partition)):
            raise ValueError("COO matvec accumulates (beta = 1)")
        touch = getattr(dst, "_touch", None)
        if touch is not None:
            touch()
        for i, c in enumerate(kernel_partition.colors):
            p = range_partition.pieces[c]
            k_lo, k_hi = kernel_partition.intervals[i]
            nnz = max(k_hi - k_lo + 1, 0)
            if p.volume == 0 or nnz == 0:
                continue
            if c in self.sliced and self.sliced_window == (x_lo, x_hi):
                self.ctx.backend.csr_sliced_spmv(self.sliced[c], 0, p.volume, x, dst._piece(i), beta,
                                                 self.sliced_placement.get(c, 0))
                continue
            if self.binned is not None and self.binned_window == (x_lo, x_hi) and c == self.binned_color:
                self.ctx.backend.coo_binned_spmv(self.binned, x, dst._piece(i), beta)
                continue
            if self.blocked and self.col32_window == (x_lo, x_hi):
                # (one local piece) every entry is visited; the row window selects, as always
                self.ctx.backend.coo_spmv(self.nnz, self.row32, self.col32, self.val_blocked, x, x_lo, x_hi,
                                          dst._piece(i), p.lo[0] - self.row_origin, p.hi[0] - self.row_origin)
                continue
            if self.col32 is not None and self.col32_window == (x_lo, x_hi):
                self.ctx.backend.coo_spmv(nnz, self.row32[k_lo:k_lo + nnz], self.col32[k_lo:k_lo + nnz],
                                          self.val[k_lo:k_lo + nnz], x, x_lo, x_hi, dst._piece(i),
                                          p.lo[0] - self.row_origin, p.hi[0] - self.row_origin)
                continue
            self.ctx.backend.coo_spmv(nnz, self.row[k_lo:k_lo + nnz], self.col[k_lo:k_lo + nnz],
                                      self.val[k_lo:k_lo + nnz], x, x_lo, x_hi, dst._piece(i),
                                      p.lo[0], p.hi[0])


def _ghost_from_kernel(ctx, col, k_base, domain_space, kernel_partition) -> PiecePartition:
    d_lo, d_hi = domain_space.rect.lo[0], domain_space.rect.hi[0]
    out = []
    for k_lo, k_hi in kernel_partition.intervals:
        if k_hi < k_lo:
            out.append((0, -1))
            continue
        lo, hi = ctx.backend.index_bounds(col[k_lo - k_base:k_hi - k_base + 1])
        lo, hi = max(lo, d_lo), min(hi, d_hi)  # the image is taken inside the domain space
        out.append((lo, hi) if hi >= lo else (0, -1))
    return PiecePartition(kernel_partition.colors, out)
