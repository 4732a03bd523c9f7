"""Shared helpers for the tests: build the same problem for the oracle planner and for the
host mirror (on whichever backend the context carries), from global numpy arrays."""
from __future__ import annotations

import numpy as np
import torch

import oracle
from oracle import partition as part
from oracle import solvers as S

from legionsolvers_amd import (COOMatrix, CSRMatrix, IndexPartition, IndexSpace, PartitionedVector,
                               SquarePlanner)

TORCH_DT = {np.dtype(np.float32): torch.float32, np.dtype(np.float64): torch.float64}


def stencil_problem(flag, extent, dtype=np.float64, coo=False):
    """(rects, row, col, val, n) of a BenchmarkStencil matrix, from the oracle generator."""
    offs, cfs = oracle.benchmark_stencil_offsets(flag)
    rects, row, col, val = oracle.linearized_stencil(extent, offs, cfs, dtype=dtype, coo=True)
    return rects, row, col, val, int(np.prod(extent))


def oracle_planner(kind, rects, row, col, val, n, pieces, dtype, rhs=None, spaces=1):
    pl = S.Planner(dtype)
    pcs = part.equal_partition(0, n - 1, pieces)
    for _ in range(spaces):
        pl.add_sol_vector(np.zeros(n, dtype), pcs)
    for _ in range(spaces):
        pl.add_rhs_vector(np.ones(n, dtype) if rhs is None else rhs.astype(dtype).copy(), pcs)
    for s in range(spaces):
        pl.add_row_partitioned_matrix(kind, rects, row, col, val, s, s)
    return pl


def to_dev(ctx, a: np.ndarray) -> torch.Tensor:
    return torch.from_numpy(np.ascontiguousarray(a)).to(ctx.device)


def shard_matrix(ctx, kind, rects, row, col, val, n, partition):
    """This rank's shard of a global matrix, as the device generators would leave it."""
    row_lo, row_hi = partition.rank_interval(ctx.rank, ctx.world_size)
    space = IndexSpace(0, n - 1)
    if kind == "csr":
        r = rects[row_lo:row_hi + 1]
        live = r[r[:, 1] >= r[:, 0]]
        k_lo, k_hi = (int(live[:, 0].min()), int(live[:, 1].max())) if len(live) else (0, -1)
        return CSRMatrix(ctx, to_dev(ctx, r), k_lo, to_dev(ctx, col[k_lo:k_hi + 1]),
                         to_dev(ctx, val[k_lo:k_hi + 1]), space, row_lo)
    sel = (row >= row_lo) & (row <= row_hi)
    return COOMatrix(ctx, to_dev(ctx, row[sel]), to_dev(ctx, col[sel]), to_dev(ctx, val[sel]), space)


def mirror_planner(ctx, kind, rects, row, col, val, n, pieces, dtype, rhs=None, spaces=1):
    tdt = TORCH_DT[np.dtype(dtype)]
    partition = IndexPartition.equal(IndexSpace(0, n - 1), pieces)
    pl = SquarePlanner(ctx, tdt)
    sols, rhss = [], []
    for s in range(spaces):
        v = PartitionedVector(ctx, f"sol{s}", partition, tdt)
        v.zero_fill()
        sols.append(v)
        pl.add_sol_vector(v)
    for s in range(spaces):
        v = PartitionedVector(ctx, f"rhs{s}", partition, tdt)
        if rhs is None:
            v.constant_fill(1.0)
        else:
            v.owned.copy_(to_dev(ctx, rhs.astype(dtype)[v.own_lo:v.own_hi + 1]))
        rhss.append(v)
        pl.add_rhs_vector(v)
    matrix = shard_matrix(ctx, kind, rects, row, col, val, n, partition)
    for s in range(spaces):
        pl.add_row_partitioned_matrix(matrix, s, s)
    return pl


def gather_vector(ctx, v: PartitionedVector) -> np.ndarray:
    """The whole vector on every rank, as numpy."""
    parts = ctx.comm.all_gather_object(v.to_host().numpy().copy())
    return np.concatenate(parts)
