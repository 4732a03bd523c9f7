"""Stencil and example-system matrix builders: the matrices the reference's tests and
BenchmarkStencil run on (src/StencilGenerator.{hpp,cpp}, src/ExampleSystems.{hpp,cpp}),
generated on the device, each rank filling only the rows it owns.

Output is bit-identical to the reference's fill tasks with IndexOrder::ROW_MAJOR: offsets
sorted lexicographically (ties by coefficient), rows numbered by row-major linearisation of the
grid, an entry wherever point+offset stays inside the grid, kernel indices counting entries in
row order, row rect = {first, last} entry of the row.
"""
from __future__ import annotations

import ctypes
from typing import Sequence, Tuple

import torch

from . import _lib
from .index_space import IndexPartition, IndexSpace
from .matrices import COOMatrix, CSRMatrix


def sort_offsets(offsets: Sequence[Sequence[int]], coeffs: Sequence[float]):
    """compare_row_major on the point, then the entry (src/StencilGenerator.cpp:40-52)."""
    pairs = sorted(zip([tuple(int(v) for v in o) for o in offsets], [float(c) for c in coeffs]))
    return [p[0] for p in pairs], [p[1] for p in pairs]


def benchmark_stencil(dim_flag: int):
    """(extent dimension, offsets, coefficients) of test/BenchmarkStencil.cpp's -dim flag:
    1: 1-D [-1 2 -1]; 2: 2-D 5-point; 3: 3-D 7-point; 4: 3-D 27-point with centre 88/26,
    faces -6/26, edges -3/26, corners -2/26 (:33-131)."""
    if dim_flag == 1:
        return 1, [(0,), (-1,), (1,)], [2.0, -1.0, -1.0]
    if dim_flag == 2:
        return 2, [(0, 0), (-1, 0), (1, 0), (0, -1), (0, 1)], [4.0, -1.0, -1.0, -1.0, -1.0]
    if dim_flag == 3:
        offs = [(0, 0, 0), (-1, 0, 0), (1, 0, 0), (0, -1, 0), (0, 1, 0), (0, 0, -1), (0, 0, 1)]
        return 3, offs, [6.0] + [-1.0] * 6
    if dim_flag == 4:
        by_weight = {0: 88.0 / 26.0, 1: -6.0 / 26.0, 2: -3.0 / 26.0, 3: -2.0 / 26.0}
        offs = [(a, b, c) for a in (-1, 0, 1) for b in (-1, 0, 1) for c in (-1, 0, 1)]
        return 3, offs, [by_weight[abs(a) + abs(b) + abs(c)] for a, b, c in offs]
    raise ValueError("INVALID DIM")  # as test/BenchmarkStencil.cpp:132-135


def _i64_array(values):
    flat = [int(v) for v in values]
    return (ctypes.c_int64 * len(flat))(*flat)


def _build(ctx, extent, offsets, coeffs, index_partition: IndexPartition, dtype, coo: bool):
    if not torch.cuda.is_available():
        raise RuntimeError("the stencil generator runs on the GPU; no CPU fallback")
    dim = len(extent)
    rows_total = 1
    for e in extent:
        rows_total *= int(e)
    if index_partition.space.dim != 1 or index_partition.space.volume != rows_total:
        raise ValueError("the vector partition must colour the linearised grid 0..rows-1")
    offs, cfs = sort_offsets(offsets, coeffs)
    ext_c = _i64_array(extent)
    off_c = _i64_array([v for o in offs for v in o])
    suf = _lib.suffix(dtype)
    cf_c = ((ctypes.c_double if suf == "f64" else ctypes.c_float) * len(cfs))(*cfs)
    be = ctx.backend
    dev = ctx.device
    row_lo, row_hi = index_partition.rank_interval(ctx.rank, ctx.world_size)
    rows = max(row_hi - row_lo + 1, 0)

    counts = torch.empty(rows, dtype=torch.int64, device=dev)
    _lib.call("lsk_stencil_count", dim, ext_c, len(offs), off_c, row_lo, row_hi,
              ctypes.c_void_p(counts.data_ptr()), be.stream())
    prefix = torch.empty_like(counts)
    _lib.call("lsk_inclusive_sum_i64", rows, ctypes.c_void_p(counts.data_ptr()),
              ctypes.c_void_p(prefix.data_ptr()), be.stream())
    nnz = int(prefix[-1].item()) if rows else 0
    k_base = sum(ctx.comm.all_gather_object(nnz)[:ctx.rank])

    rects = torch.empty((rows, 2), dtype=torch.int64, device=dev)
    _lib.call("lsk_rects_from_counts_i64", rows, ctypes.c_void_p(counts.data_ptr()),
              ctypes.c_void_p(prefix.data_ptr()), k_base, ctypes.c_void_p(rects.data_ptr()), be.stream())
    del counts, prefix
    col = torch.empty(nnz, dtype=torch.int64, device=dev)
    val = torch.empty(nnz, dtype=dtype, device=dev)
    space = IndexSpace(0, rows_total - 1)
    if coo:
        row = torch.empty(nnz, dtype=torch.int64, device=dev)
        _lib.call(f"lsk_stencil_fill_coo_{suf}", dim, ext_c, len(offs), off_c, cf_c, row_lo, row_hi,
                  ctypes.c_void_p(rects.data_ptr()), k_base, ctypes.c_void_p(row.data_ptr()),
                  ctypes.c_void_p(col.data_ptr()), ctypes.c_void_p(val.data_ptr()), be.stream())
        be.synchronize()
        return COOMatrix(ctx, row, col, val, space)
    _lib.call(f"lsk_stencil_fill_csr_{suf}", dim, ext_c, len(offs), off_c, cf_c, row_lo, row_hi,
              ctypes.c_void_p(rects.data_ptr()), k_base, ctypes.c_void_p(col.data_ptr()),
              ctypes.c_void_p(val.data_ptr()), be.stream())
    be.synchronize()  # the ctypes argument arrays must outlive the launches
    return CSRMatrix(ctx, rects, k_base, col, val, space, row_lo)


def create_linearized_csr_stencil_matrix(ctx, extent: Sequence[int], offsets, coeffs,
                                         index_partition: IndexPartition,
                                         dtype: torch.dtype = torch.float64) -> CSRMatrix:
    """create_linearized_csr_stencil_matrix (src/StencilGenerator.hpp, fill task
    src/StencilGenerator.cpp:379-517) for the grid [0, extent)."""
    return _build(ctx, tuple(extent), offsets, coeffs, index_partition, dtype, coo=False)


def create_linearized_coo_stencil_matrix(ctx, extent: Sequence[int], offsets, coeffs,
                                         index_partition: IndexPartition,
                                         dtype: torch.dtype = torch.float64) -> COOMatrix:
    """create_linearized_coo_stencil_matrix (fill task src/StencilGenerator.cpp:129-242)."""
    return _build(ctx, tuple(extent), offsets, coeffs, index_partition, dtype, coo=True)


def csr_negative_laplacian_1d(ctx, grid_size: int, index_partition: IndexPartition,
                              dtype: torch.dtype = torch.float64) -> CSRMatrix:
    """csr_negative_laplacian_1d (src/ExampleSystems.cpp:382-389, 448-466): the [-1 2 -1]
    stencil; its closed-form fill equals the generic stencil fill on a 1-D grid."""
    _, offs, cfs = benchmark_stencil(1)
    return create_linearized_csr_stencil_matrix(ctx, (grid_size,), offs, cfs, index_partition, dtype)


def coo_negative_laplacian_1d(ctx, grid_size: int, index_partition: IndexPartition,
                              dtype: torch.dtype = torch.float64) -> COOMatrix:
    """coo_negative_laplacian_1d (src/ExampleSystems.cpp:311-321)."""
    _, offs, cfs = benchmark_stencil(1)
    return create_linearized_coo_stencil_matrix(ctx, (grid_size,), offs, cfs, index_partition, dtype)
