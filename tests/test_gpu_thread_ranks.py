"""The N-rank Krylov path on real kernels with device-resident halos, on one GPU: the ranks are
threads of this process (tests/thread_ranks.py).  Every rank holds one row block of the headline
operator (27-point stencil), its products are split into the rows that need no ghost cell (phase 0,
launched while the halo is in flight) and the two edge runs (phase 1), on the sliced-pattern copy
both phases leave their shares of p.Ap, and every dot product crosses the ranks in one all-reduce.
Compared with the CPU oracle on the whole problem: fp64 CG residual histories within 1e-10."""
import numpy as np
import pytest
import torch

import host_cases as hc
import oracle
from oracle import solvers as S
from thread_ranks import run_ranks

from legionsolvers_amd import (BiCGStabSolver, CGSolver, GMRESSolver, IndexPartition, IndexSpace, PartitionedVector,
                               SquarePlanner)
from legionsolvers_amd import benchmark_stencil as bs

pytestmark = pytest.mark.gpu


def _oracle_history(n1, steps, solver_flag, pieces, dtype=np.float64):
    oracle.set_threads(8)
    rects, row, col, val, n = hc.stencil_problem(4, (n1, n1, n1), dtype)
    opl = hc.oracle_planner("csr", rects, row, col, val, n, pieces, dtype)
    solver = {1: S.CGSolver, 2: S.BiCGStabSolver}[solver_flag](opl) if solver_flag != 3 else S.GMRESSolver(opl, 10)
    for _ in range(steps):
        solver.step()
    return solver, opl.get_vector(0, 0)


@pytest.mark.parametrize("world,pieces,n1,sliced", [(8, 8, 128, "1"), (8, 8, 128, "0"), (4, 8, 96, "1"), (3, 6, 40, "1")],
                         ids=["8ranks-128cubed-sliced", "8ranks-128cubed-c32", "4ranks-8pieces-96cubed", "3ranks-6pieces-40cubed"])
def test_cg_thread_ranks_halo_split(monkeypatch, world, pieces, n1, sliced):
    monkeypatch.setenv("LSK_SLICED", sliced)
    monkeypatch.setenv("LSK_TUNE", "0")
    steps = 6
    print(f"case thread ranks CG world={world} pieces={pieces} {n1}^3 sliced={sliced}", flush=True)

    def body(ctx):
        matrix, planner, solver = bs.build_problem(ctx, 4, n1, n1, n1, pieces, 1, torch.float64, num_systems=1)
        entry = planner.row_partitioned_matrices[0]
        for _ in range(steps):
            solver.step()
        sol = planner.get_vector(0, 0).to_host().numpy().copy()
        split_phases = sorted({run[0] for runs in (entry.split or {}).values() for run in runs})
        return {
            "history": [s.get_value() for s in solver.residual_norm_squared],
            "sol": sol, "own": (planner.get_vector(0, 0).own_lo, planner.get_vector(0, 0).own_hi),
            "recvs": len(entry.plan.recvs), "sends": len(entry.plan.sends), "phases": split_phases,
            "sliced": bool(matrix.sliced), "fused_used": planner._fused_dot_used,
            "all_reduces": ctx.comm.all_reduces, "exchanges": ctx.comm.exchanges,
        }

    out = run_ranks(world, body)
    osolver, osol = _oracle_history(n1, steps, 1, pieces)
    want = [float(v) for v in osolver.residual_norm_squared]
    for r, res in enumerate(out):
        np.testing.assert_allclose(res["history"], want, rtol=1e-10, err_msg=f"rank {r}")
        assert res["history"] == out[0]["history"]          # every rank holds the same reduced scalars
        lo, hi = res["own"]
        assert np.abs(res["sol"] - osol[lo:hi + 1]).max() <= 1e-10 * np.abs(osol).max()
        # interior ranks exchange with both neighbours, and their products ran in two phases
        assert res["recvs"] == res["sends"] == (2 if 0 < r < world - 1 else 1)
        assert res["phases"] == [0, 1], f"rank {r}: halo/interior split not in force: {res['phases']}"
        assert res["sliced"] == (sliced == "1")
        if sliced == "1":
            assert res["fused_used"] >= steps - 1   # both phases left the shares of p.Ap
        assert res["exchanges"] >= steps and res["all_reduces"] >= 2 * steps


@pytest.mark.parametrize("solver_flag", [2, 3], ids=["bicgstab", "gmres10"])
def test_other_solvers_thread_ranks(monkeypatch, solver_flag):
    monkeypatch.setenv("LSK_TUNE", "0")
    world, pieces, n1 = 4, 4, 48
    steps = 5 if solver_flag == 2 else 12
    print(f"case thread ranks solver={solver_flag} world={world} {n1}^3", flush=True)

    def body(ctx):
        _, planner, solver = bs.build_problem(ctx, 4, n1, n1, n1, pieces, solver_flag, torch.float64, num_systems=1)
        for _ in range(steps):
            solver.step()
        v = planner.get_vector(0, 0)
        return v.to_host().numpy().copy(), (v.own_lo, v.own_hi)

    out = run_ranks(world, body)
    _, osol = _oracle_history(n1, steps, solver_flag, pieces)
    for sol, (lo, hi) in out:
        assert np.abs(sol - osol[lo:hi + 1]).max() <= 1e-9 * np.abs(osol).max()


@pytest.mark.parametrize("world,pieces,n1", [(4, 4, 64), (3, 6, 40)], ids=["4ranks-64cubed", "3ranks-6pieces-40cubed"])
def test_two_systems_share_a_pass_across_ranks(monkeypatch, world, pieces, n1):
    """Two systems on one matrix (the reference's BenchmarkStencil, test/BenchmarkStencil.cpp:181-199) on
    several ranks: both vectors' halos are posted, the rows that need none are multiplied for both systems
    in one pass over the values while they travel, the edge rows afterwards.  Every scalar and both
    solutions equal, bit for bit, those of one product per system (LSK_PAIR=0), which the tests above hold
    against the oracle."""
    monkeypatch.setenv("LSK_TUNE", "0")
    steps = 6

    def body(ctx):
        rows = n1 ** 3
        _, offsets, coeffs = bs.sg.benchmark_stencil(4)
        partition = IndexPartition.equal(IndexSpace(0, rows - 1), pieces)
        matrix = bs.sg.create_linearized_csr_stencil_matrix(ctx, (n1, n1, n1), offsets, coeffs, partition, torch.float64)
        planner = SquarePlanner(ctx, torch.float64)
        for s in range(2):
            sol = PartitionedVector(ctx, f"sol{s}", partition, torch.float64)
            sol.zero_fill()
            planner.add_sol_vector(sol)
        for s in range(2):
            rhs = PartitionedVector(ctx, f"rhs{s}", partition, torch.float64)
            rhs.constant_fill(1.0) if s == 0 else rhs.random_fill(7)   # two different systems
            planner.add_rhs_vector(rhs)
        for s in range(2):
            planner.add_row_partitioned_matrix(matrix, s, s)
        solver = CGSolver(planner)
        for _ in range(steps):
            solver.step()
        entry = planner.row_partitioned_matrices[0]
        return {"history": [s.get_value() for s in solver.residual_norm_squared],
                "sols": [planner.get_vector(0, s).to_host().numpy().copy() for s in range(2)],
                "paired": planner._paired_products, "phases": sorted({run[0] for runs in (entry.split or {}).values() for run in runs}),
                "fused_used": planner._fused_dot_used,
                "uniform": bool(matrix.sliced) and all(s.uniform_values for s in matrix.sliced.values())}

    print(f"case thread ranks two systems world={world} pieces={pieces} {n1}^3 paired", flush=True)
    paired = run_ranks(world, body)
    monkeypatch.setenv("LSK_PAIR", "0")
    print(f"case thread ranks two systems world={world} pieces={pieces} {n1}^3 one product per system", flush=True)
    single = run_ranks(world, body)
    for r, (a, b) in enumerate(zip(paired, single)):
        assert a["paired"] >= steps and b["paired"] == 0, f"rank {r}: {a['paired']} / {b['paired']} paired products"
        assert a["phases"] == [0, 1], f"rank {r}: halo/interior split not in force"
        assert a["history"] == b["history"], f"rank {r}: scalars differ"
        assert a["fused_used"] == b["fused_used"] and a["fused_used"] >= steps - 1
        for s in range(2):
            assert np.array_equal(a["sols"][s], b["sols"][s]), f"rank {r}: solution {s} differs"
        assert all(np.isfinite(a["history"]))
    # ... with the products waiting for their halos (no interior / edge split): the pair in one launch per piece
    monkeypatch.delenv("LSK_PAIR")
    monkeypatch.setenv("LSK_OVERLAP", "0")
    print(f"case thread ranks two systems world={world} pieces={pieces} {n1}^3 paired, no overlap", flush=True)
    waited = run_ranks(world, body)
    monkeypatch.delenv("LSK_OVERLAP")
    for r, (a, c) in enumerate(zip(paired, waited)):
        assert c["paired"] >= steps and c["phases"] == [], f"rank {r}: {c['paired']} paired products, phases {c['phases']}"
        assert a["history"] == c["history"] and all(np.array_equal(a["sols"][s], c["sols"][s]) for s in range(2)), f"rank {r}"
    monkeypatch.setenv("LSK_PAIR", "0")
    # ... and with the stencil's coefficients taken as scalars (option sliced.uniform): the same bits again
    from legionsolvers_amd import _lib
    monkeypatch.delenv("LSK_PAIR")
    _lib.call("lsk_set_option", b"sliced.uniform", 1)
    try:
        print(f"case thread ranks two systems world={world} pieces={pieces} {n1}^3 paired, scalar coefficients", flush=True)
        scalar = run_ranks(world, body)
    finally:
        _lib.call("lsk_set_option", b"sliced.uniform", 0)
    for r, (a, c) in enumerate(zip(paired, scalar)):
        assert c["uniform"], f"rank {r}: the pieces' values should be uniform per list position"
        assert a["history"] == c["history"], f"rank {r}: scalars differ with scalar coefficients"
        for s in range(2):
            assert np.array_equal(a["sols"][s], c["sols"][s]), f"rank {r}: solution {s} differs with scalar coefficients"


@pytest.mark.parametrize("world", [2, 3])
def test_unstructured_coo_starts_on_its_own_columns_across_ranks(monkeypatch, world):
    """An unstructured COO matrix reads all of x on every rank.  On the binned copy the first pass of the
    blocks of columns a rank owns is launched while the other ranks' parts of x are still on their way,
    the rest and the second pass once they are here (SquarePlanner.matvec, lsk_coo_binned_spmv_phase_*):
    BiCGStab across the ranks follows the oracle's single-process run."""
    from legionsolvers_amd import COOMatrix
    monkeypatch.setenv("LSK_TUNE", "0")
    monkeypatch.setenv("LSK_COO_BLOCK", "1")
    monkeypatch.setenv("LSK_COO_BINNED", "2")
    monkeypatch.setattr(COOMatrix, "BLOCK_MIN_WINDOW", 0)
    monkeypatch.setattr(COOMatrix, "BINNED_BLOCK_COLS", 256)
    monkeypatch.setattr(COOMatrix, "BINNED_TILE_ROWS", 512)
    rng = np.random.default_rng(83)
    n, per, steps = 24001, 12, 5   # (at least 2^16 entries per rank, or the planner leaves the row order alone)
    row = np.repeat(np.arange(n), per).astype(np.int64)
    col = rng.integers(0, n, size=n * per).astype(np.int64)
    col[::per] = np.arange(n)                      # a dominant diagonal keeps the system well posed
    val = rng.uniform(-1, 1, size=n * per)
    val[::per] = 2.0 * per
    rhs = rng.standard_normal(n)
    print(f"case thread ranks unstructured coo world={world}", flush=True)
    opl = hc.oracle_planner("coo", None, row, col, val, n, world, np.float64, rhs)
    so = S.BiCGStabSolver(opl)
    for _ in range(steps):
        so.step()
    want = opl.get_vector(0, 0)

    def body(ctx):
        mpl = hc.mirror_planner(ctx, "coo", None, row, col, val, n, world, np.float64, rhs)
        matrix = mpl.row_partitioned_matrices[0].matrix
        assert matrix.binned is not None
        sm = BiCGStabSolver(mpl)
        for _ in range(steps):
            sm.step()
        v = mpl.get_vector(0, 0)
        return v.to_host().numpy().copy(), (v.own_lo, v.own_hi), getattr(mpl, "coo_products_in_phases", 0)

    for sol, (lo, hi), phased in run_ranks(world, body):
        assert phased >= 2 * steps, phased          # both products of every step went in two phases
        assert np.abs(sol - want[lo:hi + 1]).max() <= 1e-9 * np.abs(want).max()
