"""Index spaces and partitions: the small part of Legion's index-space tree the Krylov path
uses.  Pure host arithmetic, no GPU.

A vector lives on a dense N-D rect (`IndexSpace`); an `IndexPartition` colours it with
disjoint sub-rects, one per colour of a 1-D colour space, as `create_equal_partition` does in
the reference's programs (test/BenchmarkStencil.cpp:175-179, test/Test02VectorOperations.cpp:
105-115).  Colours are dealt to ranks in contiguous blocks, the placement the reference's
BlockingShardingFunctor gives point tasks (src/LegionSolversMapper.cpp:140-151).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List, Sequence, Tuple


@dataclass(frozen=True)
class Rect:
    """Dense N-D rectangle with inclusive bounds (Legion::Rect<DIM>)."""
    lo: Tuple[int, ...]
    hi: Tuple[int, ...]

    def __post_init__(self):
        if len(self.lo) != len(self.hi) or not 1 <= len(self.lo) <= 3:
            raise ValueError("Rect needs matching lo/hi of dimension 1..3")

    @staticmethod
    def make(lo, hi) -> "Rect":
        if isinstance(lo, int):
            lo, hi = (lo,), (hi,)
        return Rect(tuple(int(v) for v in lo), tuple(int(v) for v in hi))

    @property
    def dim(self) -> int:
        return len(self.lo)

    @property
    def extent(self) -> Tuple[int, ...]:
        return tuple(max(h - l + 1, 0) for l, h in zip(self.lo, self.hi))

    @property
    def volume(self) -> int:
        v = 1
        for e in self.extent:
            v *= e
        return v

    @property
    def empty(self) -> bool:
        return self.volume == 0


class IndexSpace:
    """A dense index space (the reference's vectors never use sparse spaces)."""

    def __init__(self, lo, hi=None):
        self.rect = lo if isinstance(lo, Rect) else Rect.make(lo, hi)

    @property
    def dim(self) -> int:
        return self.rect.dim

    @property
    def volume(self) -> int:
        return self.rect.volume

    def __eq__(self, other):
        return isinstance(other, IndexSpace) and self.rect == other.rect

    def __hash__(self):
        return hash(self.rect)

    def __repr__(self):
        return f"IndexSpace({self.rect.lo}..{self.rect.hi})"


def split_equal(lo: int, hi: int, pieces: int) -> List[Tuple[int, int]]:
    """Bounds of `pieces` consecutive blocks of lo..hi: block i takes
    floor(remaining / (pieces - i)) points, so sizes differ by at most one and the larger
    blocks come last.  Empty blocks are (p, p - 1)."""
    if pieces < 1:
        raise ValueError("a partition needs at least one colour")
    out = []
    total = max(hi - lo + 1, 0)
    px = lo
    for i in range(pieces):
        nx = (total - (px - lo)) // (pieces - i)
        out.append((px, px + nx - 1))
        px += nx
    return out


class IndexPartition:
    """Disjoint, complete colouring of an index space by sub-rects, colour c -> pieces[c]."""

    def __init__(self, space: IndexSpace, pieces: Sequence[Rect]):
        self.space = space
        self.pieces = list(pieces)
        if sum(p.volume for p in self.pieces) != space.volume:
            raise ValueError("partition is not complete and disjoint")

    @staticmethod
    def equal(space: IndexSpace, num_colors: int) -> "IndexPartition":
        """create_equal_partition: split the leading dimension into num_colors blocks."""
        r = space.rect
        pieces = []
        for lo0, hi0 in split_equal(r.lo[0], r.hi[0], num_colors):
            pieces.append(Rect((lo0,) + r.lo[1:], (hi0,) + r.hi[1:]))
        return IndexPartition(space, pieces)

    @property
    def num_colors(self) -> int:
        return len(self.pieces)

    def __eq__(self, other):
        return (isinstance(other, IndexPartition) and self.space == other.space
                and self.pieces == other.pieces)

    def __hash__(self):
        return hash((self.space, tuple(self.pieces)))

    # -- placement --------------------------------------------------------------------
    def colors_of_rank(self, rank: int, world_size: int) -> range:
        """Colours are dealt to ranks in contiguous blocks of num_colors / world_size."""
        if self.num_colors % world_size != 0:
            raise ValueError(
                f"{self.num_colors} pieces do not divide over {world_size} ranks")
        per = self.num_colors // world_size
        return range(rank * per, (rank + 1) * per)

    def rank_interval(self, rank: int, world_size: int) -> Tuple[int, int]:
        """1-D spaces: the contiguous interval (lo, hi) a rank owns ((p, p-1) if empty)."""
        if self.space.dim != 1:
            raise ValueError("rank intervals exist for 1-D spaces only")
        colors = self.colors_of_rank(rank, world_size)
        first, last = self.pieces[colors[0]], self.pieces[colors[-1]]
        return first.lo[0], last.hi[0]
