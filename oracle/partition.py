"""Restatement of the partition arithmetic the reference delegates to Legion's dependent
partitioning, for 1-D index spaces.  TEST INFRASTRUCTURE (see oracle/__init__.py).

Pieces are inclusive (lo, hi) pairs or sorted index arrays.
"""
from __future__ import annotations

import numpy as np


def equal_partition(lo: int, hi: int, pieces: int):
    """Runtime::create_equal_partition over the 1-D space lo..hi (test/BenchmarkStencil.cpp:
    178-179, test/Test06CSRSolveCG.cpp:59-60).  Realm splits the leading dimension so that
    piece i takes floor(remaining / (pieces - i)) points.  The reference only ever uses sizes
    divisible by the piece count, where every rule gives equal blocks; the rule for a
    remainder is from memory of Realm's source and is not pinned by any reference golden."""
    out = []
    total = max(hi - lo + 1, 0)
    px = lo
    for i in range(pieces):
        nx = (total - (px - lo)) // (pieces - i)
        out.append((px, px + nx - 1))
        px += nx
    return out


def image_range(rects: np.ndarray, row_piece):
    """create_partition_by_image_range over the rowptr field (src/CSRMatrix.cpp:89-109): the
    kernel indices covered by the rects of the rows lo..hi of a range piece.  `rects` is the
    global (rows, 2) array.  Returns a sorted index array."""
    lo, hi = row_piece
    if hi < lo:
        return np.zeros(0, dtype=np.int64)
    r = rects[lo:hi + 1]
    r = r[r[:, 1] >= r[:, 0]]
    if len(r) == 0:
        return np.zeros(0, dtype=np.int64)
    if np.all(r[1:, 0] == r[:-1, 1] + 1):  # back-to-back rects cover one interval
        return np.arange(r[0, 0], r[-1, 1] + 1, dtype=np.int64)
    return np.unique(np.concatenate([np.arange(a, b + 1, dtype=np.int64) for a, b in r]))


def preimage(field: np.ndarray, piece):
    """create_partition_by_preimage over a row (or column) field (src/COOMatrix.cpp:77-95,
    src/CSRMatrix.cpp:68-86): kernel indices k with field[k] inside the piece."""
    lo, hi = piece
    return np.nonzero((field >= lo) & (field <= hi))[0].astype(np.int64)


def image(field: np.ndarray, kernel_piece: np.ndarray, space=None):
    """create_partition_by_image over the column field (src/CSRMatrix.cpp:112-132): the set
    of field values at the kernel indices of a piece, clipped to `space` = (lo, hi)."""
    vals = np.unique(field[kernel_piece])
    if space is not None:
        vals = vals[(vals >= space[0]) & (vals <= space[1])]
    return vals.astype(np.int64)


def image_bounds(field: np.ndarray, kernel_piece: np.ndarray, space=None):
    """bounds(image(field, kernel_piece, space)) without building the set."""
    vals = field[kernel_piece]
    if space is not None:
        vals = vals[(vals >= space[0]) & (vals <= space[1])]
    return bounds(vals)


def bounds(indices: np.ndarray):
    """Bounding interval of an index set ((0, -1) if empty): what a physical instance of the
    piece covers and what the leaf tasks see as `input_domain.bounds`."""
    if indices.size == 0:
        return (0, -1)
    return (int(indices.min()), int(indices.max()))


def partition_lines(name: str, pieces) -> list:
    """What print_index_partition + PrintIndexTask write for a 1-D partition
    (src/LegionUtilities.cpp:24-69, src/UtilityTasks.cpp:102-125), one line per point:
    `<name> (<colour>) <<point>>`."""
    lines = []
    for color, piece in enumerate(pieces):
        idx = np.arange(piece[0], piece[1] + 1) if isinstance(piece, tuple) else piece
        for i in idx:
            lines.append(f"{name} ({color}) <{int(i)}>")
    return lines
