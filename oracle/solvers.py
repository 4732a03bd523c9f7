"""Restatement of the reference's planner and solver sequences on top of the oracle leaf ops.
TEST INFRASTRUCTURE (see oracle/__init__.py).

Vectors are full-length numpy arrays; every vector operation is applied piece by piece, as
the reference's index launches do, and a dot product is the per-piece results folded in colour
order.  Scalars are numpy scalars of the entry type and every scalar operation goes through
oracle.scalar_op so that fp32 runs stay in fp32.
"""
from __future__ import annotations

import numpy as np

import oracle
from oracle import partition as part


class Matrix:
    """A CSR or COO matrix with its row-partitioned pieces (SquarePlanner.hpp:209-235)."""

    def __init__(self, kind, rects, row, col, val, range_pieces, domain_space):
        self.kind = kind
        self.rects, self.row, self.col, self.val = rects, row, col, val
        self.range_pieces = range_pieces
        # kernel partition from the range partition, then ghost (domain) partition from it
        if kind == "csr":
            self.kernel_pieces = [part.image_range(rects, p) for p in range_pieces]
        else:
            self.kernel_pieces = [part.preimage(row, p) for p in range_pieces]
        self.ghost_bounds = [part.image_bounds(col, k, domain_space) for k in self.kernel_pieces]

    def matvec(self, y, x):
        """Index launch of CSRMatvecTask / COOMatvecTask, CPU bodies, accumulate."""
        for color, (rlo, rhi) in enumerate(self.range_pieces):
            kp = self.kernel_pieces[color]
            if rhi < rlo:
                continue
            klo, khi = part.bounds(kp)
            glo, ghi = self.ghost_bounds[color]
            xin = np.ascontiguousarray(x[glo:ghi + 1])
            yp = np.ascontiguousarray(y[rlo:rhi + 1])
            if khi >= klo:
                ksl = slice(klo, khi + 1)
                if self.kind == "csr":
                    oracle.csr_matvec(np.ascontiguousarray(self.rects[rlo:rhi + 1]), klo,
                                      np.ascontiguousarray(self.col[ksl]),
                                      np.ascontiguousarray(self.val[ksl]), xin, glo, ghi, yp, beta=1)
                else:
                    oracle.coo_matvec(np.ascontiguousarray(self.row[ksl]),
                                      np.ascontiguousarray(self.col[ksl]),
                                      np.ascontiguousarray(self.val[ksl]), xin, glo, ghi, yp, rlo, rhi)
            y[rlo:rhi + 1] = yp


class Planner:
    """SquarePlanner (src/SquarePlanner.hpp): vector index 0 = SOL, 1 = RHS, 2.. = workspace."""

    def __init__(self, dtype):
        self.dtype = np.dtype(dtype)
        self.pieces = []      # per space: list of (lo, hi)
        self.sol, self.rhs, self.work = [], [], []
        self.matrices = []    # (Matrix, domain_index, range_index)

    # -- construction -----------------------------------------------------------------
    def add_sol_vector(self, v, pieces):
        self.sol.append(v)
        if len(self.pieces) < len(self.sol):
            self.pieces.append(pieces)
        return len(self.sol) - 1

    def add_rhs_vector(self, v, pieces):
        self.rhs.append(v)
        if len(self.pieces) < len(self.rhs):
            self.pieces.append(pieces)
        return len(self.rhs) - 1

    def add_row_partitioned_matrix(self, kind, rects, row, col, val, domain_index, range_index):
        dom = (0, self.sol[domain_index].size - 1)
        self.matrices.append((Matrix(kind, rects, row, col, val, self.pieces[range_index], dom),
                              domain_index, range_index))

    def allocate_workspace(self, n):
        self.work = [[np.zeros(s.size, dtype=self.dtype) for s in self.sol] for _ in range(n)]

    def num_spaces(self):
        return len(self.sol)

    def get_vector(self, idx, space):
        if idx == 0:
            return self.sol[space]
        if idx == 1:
            return self.rhs[space]
        return self.work[idx - 2][space]

    def scalar(self, value):
        return self.dtype.type(value)

    # -- vector operations (SquarePlanner.hpp:248-357) ---------------------------------
    def _each_piece(self, space):
        for lo, hi in self.pieces[space]:
            if hi >= lo:
                yield slice(lo, hi + 1)

    def zero_fill(self, idx):
        for s in range(self.num_spaces()):
            self.get_vector(idx, s)[:] = 0

    def copy(self, dst, src):
        for s in range(self.num_spaces()):
            self.get_vector(dst, s)[:] = self.get_vector(src, s)

    def scal(self, dst, *futures):
        alpha = oracle.get_alpha(futures, self.dtype)
        for s in range(self.num_spaces()):
            v = self.get_vector(dst, s)
            for sl in self._each_piece(s):
                piece = np.ascontiguousarray(v[sl])
                oracle.scal(alpha, piece)
                v[sl] = piece

    def axpy(self, dst, *args):
        *futures, src = args
        alpha = oracle.get_alpha(futures, self.dtype)
        for s in range(self.num_spaces()):
            y, x = self.get_vector(dst, s), self.get_vector(src, s)
            for sl in self._each_piece(s):
                piece = np.ascontiguousarray(y[sl])
                oracle.axpy(alpha, np.ascontiguousarray(x[sl]), piece)
                y[sl] = piece

    def xpay(self, dst, *args):
        *futures, src = args
        alpha = oracle.get_alpha(futures, self.dtype)
        for s in range(self.num_spaces()):
            y, x = self.get_vector(dst, s), self.get_vector(src, s)
            for sl in self._each_piece(s):
                piece = np.ascontiguousarray(y[sl])
                oracle.xpay(alpha, np.ascontiguousarray(x[sl]), piece)
                y[sl] = piece

    def _dot_space(self, v_idx, w_idx, s):
        v, w = self.get_vector(v_idx, s), self.get_vector(w_idx, s)
        result = None
        for lo, hi in self.pieces[s]:  # empty pieces return 0 and still take part in the fold
            d = oracle.dot(np.ascontiguousarray(v[lo:hi + 1]), np.ascontiguousarray(w[lo:hi + 1])) \
                if hi >= lo else self.dtype.type(0)
            result = d if result is None else oracle.scalar_op("add", result, d, self.dtype)
        return result

    def dot(self, v_idx, w_idx):
        result = self._dot_space(v_idx, w_idx, 0)
        for s in range(1, self.num_spaces()):
            result = oracle.scalar_op("add", result, self._dot_space(v_idx, w_idx, s), self.dtype)
        return result

    def matvec(self, dst, src):
        self.zero_fill(dst)
        for m, dom, rng in self.matrices:
            m.matvec(self.get_vector(dst, rng), self.get_vector(src, dom))


SOL, RHS = 0, 1


class CGSolver:
    """src/CGSolver.hpp:32-55."""
    P, Q, R = 2, 3, 4

    def __init__(self, planner):
        self.planner = pl = planner
        self.negative_one = pl.scalar(-1)
        pl.allocate_workspace(3)
        pl.copy(self.P, RHS)
        pl.copy(self.R, RHS)
        self.residual_norm_squared = [pl.dot(self.R, self.R)]

    def step(self):
        pl = self.planner
        pl.matvec(self.Q, self.P)
        p_norm = pl.dot(self.P, self.Q)
        r_norm2_old = self.residual_norm_squared[-1]
        pl.axpy(SOL, r_norm2_old, p_norm, self.P)
        pl.axpy(self.R, self.negative_one, r_norm2_old, p_norm, self.Q)
        r_norm2_new = pl.dot(self.R, self.R)
        self.residual_norm_squared.append(r_norm2_new)
        pl.xpay(self.P, r_norm2_new, r_norm2_old, self.R)


class BiCGStabSolver:
    """src/BiCGStabSolver.hpp:36-82."""
    P, R, R_TILDE, U, V = 2, 3, 4, 5, 6

    def __init__(self, planner):
        self.planner = pl = planner
        self.dt = pl.dtype
        self.negative_one, self.zero, self.one = pl.scalar(-1), pl.scalar(0), pl.scalar(1)
        pl.allocate_workspace(5)
        pl.copy(self.R, RHS)
        pl.copy(self.R_TILDE, RHS)
        self.rho, self.alpha, self.omega = [self.one], [self.zero], [self.one]
        pl.zero_fill(self.P)
        pl.zero_fill(self.V)

    def _op(self, op, a, b=0.0):
        return oracle.scalar_op(op, a, b, self.dt)

    def step(self):
        pl = self.planner
        rho_new = pl.dot(self.R, self.R_TILDE)
        beta = self._op("mul", self._op("div", rho_new, self.rho[-1]),
                        self._op("div", self.alpha[-1], self.omega[-1]))
        self.rho.append(rho_new)
        pl.axpy(self.P, self._op("neg", self.omega[-1]), self.V)
        pl.xpay(self.P, beta, self.R)
        pl.matvec(self.V, self.P)
        temp = pl.dot(self.R_TILDE, self.V)
        pl.axpy(self.R, self.negative_one, self.rho[-1], temp, self.V)
        self.alpha.append(self._op("div", self.rho[-1], temp))
        pl.matvec(self.U, self.R)
        r_anorm2 = pl.dot(self.R, self.U)
        u_norm2 = pl.dot(self.U, self.U)
        self.omega.append(self._op("div", r_anorm2, u_norm2))
        pl.axpy(SOL, self.alpha[-1], self.P)
        pl.axpy(SOL, self.omega[-1], self.R)
        pl.axpy(self.R, self._op("neg", self.omega[-1]), self.U)


class GMRESSolver:
    """src/GMRESSolver.hpp:32-127, including the unfinished least-squares solve: the update
    coefficient is DummyTask's constant 1 (:109-126, src/UtilityTasks.cpp:96-99)."""

    def __init__(self, planner, restart, orthogonalization="modified"):
        """orthogonalization "classical": every h_k = v_k . w of a sweep from the SAME w, then the
        updates w <- fma(-h_k, v_k, w) in ascending k; no counterpart in the reference, which runs the
        modified sweep (src/GMRESSolver.hpp:92-100); restated here as the checker of the planner's
        multi_dot / multi_axpy."""
        assert orthogonalization in ("modified", "classical")
        self.orthogonalization = orthogonalization
        self.planner = pl = planner
        self.restart = restart
        self.dt = pl.dtype
        self.negative_one = pl.scalar(-1)
        pl.allocate_workspace(restart + 1)
        self.inner_products = [[pl.scalar(0) for _ in range(restart)] for _ in range(restart + 1)]

    @staticmethod
    def krylov_basis(i):
        return i + 2

    def _op(self, op, a, b=0.0):
        return oracle.scalar_op(op, a, b, self.dt)

    def step(self):
        pl, kb = self.planner, self.krylov_basis
        pl.matvec(kb(0), SOL)
        pl.xpay(kb(0), self.negative_one, RHS)
        residual_norm = self._op("rsqrt", pl.dot(kb(0), kb(0)))
        pl.scal(kb(0), residual_norm)
        for j in range(self.restart):
            pl.matvec(kb(j + 1), kb(j))
            if self.orthogonalization == "classical":
                for k in range(j + 1):
                    self.inner_products[k][j] = pl.dot(kb(k), kb(j + 1))
                for k in range(j + 1):
                    pl.axpy(kb(j + 1), self._op("neg", self.inner_products[k][j]), kb(k))
            else:
                for k in range(j + 1):
                    self.inner_products[k][j] = pl.dot(kb(k), kb(j + 1))
                    pl.axpy(kb(j + 1), self._op("neg", self.inner_products[k][j]), kb(k))
            d = pl.dot(kb(j + 1), kb(j + 1))
            self.inner_products[j + 1][j] = self._op("sqrt", d)
            if j + 1 < self.restart:
                pl.scal(kb(j + 1), self._op("rsqrt", d))
        coeff = pl.scalar(1)
        for j in range(self.restart):
            pl.axpy(SOL, coeff, kb(j))
