"""The reference's Krylov solvers, unchanged as callers of the planner: CGSolver
(src/CGSolver.hpp), BiCGStabSolver (src/BiCGStabSolver.hpp), GMRESSolver
(src/GMRESSolver.hpp).  Each step() issues exactly the reference's sequence of planner
operations; all scalars are device-resident futures, so a step enqueues without a host sync.
"""
from __future__ import annotations

from typing import List, Sequence

import torch

from .scalar import Scalar
from .square_planner import SquarePlanner

SOL = 0
RHS = 1


class CapturedStep:
    """`steps` step()s of a solver recorded once as a HIP graph; every replay() is those steps again,
    enqueued with one call.  What rt->begin_trace / rt->end_trace around the steps are in the
    reference's benchmark (test/BenchmarkStencil.cpp:219-222: the first pass records the task launches,
    later passes replay them), possible here because every scalar of a step lives on the device.
    Recording executes nothing.  A third of the host time per step on small systems, where the host is
    the limit; next to nothing where the GPU is.

    A step reads scalars the step before made (CG: r.r; BiCGStab: rho, alpha, omega) through the
    tensors of that step; the recording ends by copying the newest values into the tensors its first
    step read, so each replay continues from the one before.  The solver's history lists grow by
    `steps` entries when the steps are recorded; after every replay the last entry holds the newest
    value, and so does the entry the recording started from (the history of a replayed run is not
    kept).  The recording starts and ends at SquarePlanner.settle(): no fused leaf reaches across its
    ends.  One rank only (the exchanges of a multi-rank step are not recorded); replays go to the
    device's current stream; run a few steps before recording (a kernel's first launch on a device
    is not recordable), and leave the vectors, the matrix copies and the solver as they were for as
    long as the recording is replayed."""

    def __init__(self, solver, carried: Sequence[List[Scalar]], steps: int = 1):
        planner = solver.planner
        ctx = planner.get_context()
        if ctx.world_size != 1:
            raise NotImplementedError("steps can be recorded on one rank only")
        if ctx.device.type != "cuda":
            raise NotImplementedError("steps can be recorded on a GPU only")
        if steps < 1:
            raise ValueError("at least one step")
        import ctypes
        from . import _lib
        on = ctypes.c_int(0)
        _lib.call("lsk_get_option", b"debug.sync", ctypes.byref(on))
        if on.value:
            raise RuntimeError("option debug.sync is on: every launch waits for the device, which a recording cannot hold")
        self.solver, self.steps = solver, int(steps)
        planner.settle()
        ctx.synchronize()
        before = [scalars[-1] for scalars in carried]
        self.graph = torch.cuda.CUDAGraph()
        side = torch.cuda.Stream(ctx.device)   # (capture needs a stream of its own; default priority)
        with torch.cuda.graph(self.graph, stream=side):
            for _ in range(self.steps):
                solver.step()
            planner.settle()
            for old, scalars in zip(before, carried):
                ctx.backend.copy(old.get_future(), scalars[-1].get_future())
        self.replays = 0
        # (the recording refers to the set-up-time copies its products read: while it lives, no allocation
        # that finds no room may have them given up)
        self._backend = ctx.backend
        self._backend.recordings_alive = getattr(self._backend, "recordings_alive", 0) + 1

    def __del__(self):
        backend = getattr(self, "_backend", None)
        if backend is not None:
            backend.recordings_alive = max(getattr(backend, "recordings_alive", 1) - 1, 0)

    def replay(self, times: int = 1) -> None:
        for _ in range(times):
            self.graph.replay()
        self.replays += times


class CGSolver:
    """src/CGSolver.hpp:13-57."""
    P, Q, R = 2, 3, 4

    def __init__(self, planner: SquarePlanner):
        self.planner = planner
        self.residual_norm_squared: List[Scalar] = []
        self.negative_one = Scalar(planner.get_context(), -1.0, planner.dtype)
        planner.allocate_workspace(3)
        planner.copy(self.P, RHS)
        planner.copy(self.R, RHS)
        self.residual_norm_squared.append(planner.dot(self.R, self.R))

    def step(self) -> None:
        pl = self.planner
        pl.matvec(self.Q, self.P)
        p_norm = pl.dot(self.P, self.Q)
        r_norm2_old = self.residual_norm_squared[-1]
        pl.axpy(SOL, r_norm2_old, p_norm, self.P)
        pl.axpy(self.R, self.negative_one, r_norm2_old, p_norm, self.Q)
        r_norm2_new = pl.dot(self.R, self.R)
        self.residual_norm_squared.append(r_norm2_new)
        pl.xpay(self.P, r_norm2_new, r_norm2_old, self.R)

    def capture(self, steps: int = 1) -> CapturedStep:
        """The next `steps` steps as a replayable HIP graph (CapturedStep); run a few steps first, so
        that everything a first launch sets up is in place."""
        return CapturedStep(self, [self.residual_norm_squared], steps)


class BiCGStabSolver:
    """src/BiCGStabSolver.hpp:13-84."""
    P, R, R_TILDE, U, V = 2, 3, 4, 5, 6

    def __init__(self, planner: SquarePlanner):
        self.planner = planner
        ctx, dt = planner.get_context(), planner.dtype
        self.negative_one = Scalar(ctx, -1.0, dt)
        self.zero = Scalar(ctx, 0.0, dt)
        self.one = Scalar(ctx, 1.0, dt)
        planner.allocate_workspace(5)
        planner.copy(self.R, RHS)
        planner.copy(self.R_TILDE, RHS)
        self.rho: List[Scalar] = [self.one]
        self.alpha: List[Scalar] = [self.zero]
        self.omega: List[Scalar] = [self.one]
        planner.zero_fill(self.P)
        planner.zero_fill(self.V)

    def step(self) -> None:
        pl = self.planner
        rho_new = pl.dot(self.R, self.R_TILDE)
        beta = (rho_new / self.rho[-1]) * (self.alpha[-1] / self.omega[-1])
        self.rho.append(rho_new)
        pl.axpy(self.P, -self.omega[-1], self.V)
        pl.xpay(self.P, beta, self.R)
        pl.matvec(self.V, self.P)
        temp = pl.dot(self.R_TILDE, self.V)
        pl.axpy(self.R, self.negative_one, self.rho[-1], temp, self.V)
        self.alpha.append(self.rho[-1] / temp)
        pl.matvec(self.U, self.R)
        r_anorm2 = pl.dot(self.R, self.U)
        u_norm2 = pl.dot(self.U, self.U)
        self.omega.append(r_anorm2 / u_norm2)
        pl.axpy(SOL, self.alpha[-1], self.P)
        pl.axpy(SOL, self.omega[-1], self.R)
        pl.axpy(self.R, -self.omega[-1], self.U)

    def capture(self, steps: int = 1) -> CapturedStep:
        """The next `steps` steps as a replayable HIP graph (CapturedStep); run a few steps first."""
        return CapturedStep(self, [self.rho, self.alpha, self.omega], steps)


class GMRESSolver:
    """src/GMRESSolver.hpp:15-129.  Like the reference, the least-squares solve is not
    there yet: the solution update uses DummyTask's coefficient 1 for every basis vector
    (:109-126, src/UtilityTasks.cpp:96-99)."""

    def __init__(self, planner: SquarePlanner, restart: int, orthogonalization: str = "modified"):
        """orthogonalization: "modified" is the reference's Gram-Schmidt sweep (src/GMRESSolver.hpp:
        92-100: h_k = v_k . w and w <- w - h_k v_k alternating); "classical" forms every h_k from the
        same w, which lets all of them share one pass over w and all updates another
        (SquarePlanner.multi_dot / multi_axpy): about half the vector traffic of a restart cycle, one
        all-reduce per Krylov vector instead of one per inner product, and DIFFERENT arithmetic (less
        robust against loss of orthogonality) -- an option, never the default."""
        if orthogonalization not in ("modified", "classical"):
            raise ValueError("orthogonalization must be 'modified' or 'classical'")
        self.orthogonalization = orthogonalization
        self.planner = planner
        self.restart = int(restart)
        ctx, dt = planner.get_context(), planner.dtype
        self.negative_one = Scalar(ctx, -1.0, dt)
        planner.allocate_workspace(self.restart + 1)
        zero = Scalar(ctx, 0.0, dt)
        self.inner_products = [[zero for _ in range(self.restart)] for _ in range(self.restart + 1)]
        self.dummy_coefficient = Scalar(ctx, 1.0, dt)

    @staticmethod
    def krylov_basis(i: int) -> int:
        return i + 2

    def step(self) -> None:
        pl, kb = self.planner, self.krylov_basis
        # compute residual vector
        pl.matvec(kb(0), SOL)
        pl.xpay(kb(0), self.negative_one, RHS)
        residual_norm = pl.dot(kb(0), kb(0)).rsqrt()
        pl.scal(kb(0), residual_norm)
        for j in range(self.restart):
            pl.matvec(kb(j + 1), kb(j))
            if self.orthogonalization == "classical":
                # (in groups of what one multi_dot / multi_axpy call takes: every inner product is still
                # taken with the same w, and the updates keep their ascending order, so the bits do not
                # depend on the grouping)
                basis = [kb(k) for k in range(j + 1)]
                group = pl.get_context().backend.MULTI_MAX
                hs = [pl.multi_dot(basis[g:g + group], kb(j + 1)) for g in range(0, j + 1, group)]
                for k in range(j + 1):
                    h = hs[k // group]
                    self.inner_products[k][j] = Scalar(pl.get_context(), h[k % group:k % group + 1])
                for g, h in zip(range(0, j + 1, group), hs):
                    pl.multi_axpy(kb(j + 1), h, basis[g:g + group], negate=True)
            else:
                for k in range(j + 1):
                    self.inner_products[k][j] = pl.dot(kb(k), kb(j + 1))
                    pl.axpy(kb(j + 1), -self.inner_products[k][j], kb(k))
            d = pl.dot(kb(j + 1), kb(j + 1))
            self.inner_products[j + 1][j] = d.sqrt()
            if j + 1 < self.restart:
                pl.scal(kb(j + 1), d.rsqrt())
        coeff = self.dummy_coefficient
        for j in range(self.restart):
            pl.axpy(SOL, coeff, kb(j))

    def capture(self, steps: int = 1) -> CapturedStep:
        """The next `steps` restart cycles as a replayable HIP graph (CapturedStep): a cycle reads no
        scalar of the one before; run one cycle first."""
        return CapturedStep(self, [], steps)
