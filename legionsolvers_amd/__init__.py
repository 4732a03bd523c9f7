"""legionsolvers_amd: an MI355X-native Krylov inner loop (CSR/COO SpMV + BLAS-1) behind the
interface of dzhang314/LegionSolvers' solvers, planner, vectors and matrices.

Layers, bottom up:
  csrc/*.hip + include/lsk.h   hand-written gfx950 kernels behind a C ABI (liblsk.so)
  backend.HipBackend           one C-ABI call per leaf operation, on torch memory and streams
  Scalar, PartitionedVector, CSRMatrix, COOMatrix, SquarePlanner,
  CGSolver, BiCGStabSolver, GMRESSolver   the reference's classes, same names and call shapes
  comm.Communicator            dot all-reduce and halo gather over torch.distributed (RCCL)
"""
from .context import Context
from .index_space import IndexPartition, IndexSpace, Rect
from .matrices import AbstractMatrix, COOMatrix, CSRMatrix
from .partitioned_vector import PartitionedVector
from .scalar import Scalar
from .solvers import BiCGStabSolver, CapturedStep, CGSolver, GMRESSolver
from .square_planner import SquarePlanner

__all__ = [
    "Context", "IndexPartition", "IndexSpace", "Rect", "AbstractMatrix", "COOMatrix", "CSRMatrix",
    "PartitionedVector", "Scalar", "BiCGStabSolver", "CapturedStep", "CGSolver", "GMRESSolver", "SquarePlanner",
]
