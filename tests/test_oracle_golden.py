"""Pins the CPU oracle: every golden the reference's own tests hold for the Krylov path, the
outputs of the reference's scripts/krylov.py, and scipy.sparse cross-checks for what the
reference does not pin (2-D/3-D stencils, fp32, COO entry order)."""
import numpy as np
import pytest
import scipy.sparse as sp

import oracle
from oracle import partition as part
from oracle import solvers as S


def _cg_history(kind, n, pieces, its, dtype=np.float64, spaces=1):
    rects, row, col, val = oracle.laplacian_1d(n, dtype)
    pl = S.Planner(dtype)
    pcs = part.equal_partition(0, n - 1, pieces)
    for _ in range(spaces):
        pl.add_sol_vector(np.zeros(n, dtype), pcs)
    for _ in range(spaces):
        pl.add_rhs_vector(np.ones(n, dtype), pcs)
    for s in range(spaces):
        pl.add_row_partitioned_matrix(kind, rects, row, col, val, s, s)
    solver = S.CGSolver(pl)
    for _ in range(its):
        solver.step()
    return [float(v) for v in solver.residual_norm_squared], pl


@pytest.mark.parametrize("kind", ["coo", "csr"])
def test_cg_matches_reference_golden(kind, goldens, krylov_golden):
    """Test05COOSolveCG / Test06CSRSolveCG: stdout compared as a sorted set (test_all.py:130-133);
    krylov.py gives the order."""
    prob = goldens["cg_problem"]
    hist, pl = _cg_history(kind, prob["n"], prob["pieces"], prob["iterations"])
    printed = sorted(f"{v:g}" for v in hist)  # operator<< of a double, then the driver's sort
    assert printed == sorted(goldens["expected_cg_output_sorted_set"])
    case = krylov_golden["cases"][0]
    assert hist == case["residual_norm_squared"]  # exact: every value is an integer < 2^53
    np.testing.assert_allclose(pl.sol[0], case["x_final"], rtol=1e-12)


def test_cg_small_matches_krylov_script(krylov_golden):
    case = krylov_golden["cases"][1]
    hist, pl = _cg_history("csr", case["n"], 4, case["iterations"])
    np.testing.assert_allclose(hist, case["residual_norm_squared"], rtol=1e-12)
    np.testing.assert_allclose(pl.sol[0], case["x_final"], rtol=1e-12)


def test_two_system_planner_doubles_the_dot(goldens):
    """BenchmarkStencil registers two identical systems (test/BenchmarkStencil.cpp:201-207);
    every planner dot is then the sum over both spaces."""
    one, _ = _cg_history("csr", 100, 4, 5)
    two, _ = _cg_history("csr", 100, 4, 5, spaces=2)
    assert two == [2 * v for v in one]


def test_partition_output_matches_reference_golden(goldens):
    """Test03COOPartitioning / Test04CSRPartitioning print-out (test_all.py:19-127)."""
    n, pieces = goldens["partition_problem"]["n"], goldens["partition_problem"]["pieces"]
    rects, row, col, _ = oracle.laplacian_1d(n)
    range_pieces = part.equal_partition(0, n - 1, pieces)
    for kernel_pieces in ([part.image_range(rects, p) for p in range_pieces],   # Test04
                          [part.preimage(row, p) for p in range_pieces]):       # Test03
        domain_pieces = [part.image(col, k, (0, n - 1)) for k in kernel_pieces]
        lines = [f"[LegionSolvers] Printing index partition {name} with {pieces} pieces."
                 for name in ("domain_partition", "matrix_partition", "range_partition")]
        lines += part.partition_lines("domain_partition", domain_pieces)
        lines += part.partition_lines("matrix_partition", kernel_pieces)
        lines += part.partition_lines("range_partition", range_pieces)
        assert sorted(lines) == sorted(goldens["expected_partition_output"])


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("shape", [(100,), (10, 10), (4, 5, 5)])
def test_vector_operations_test02(dtype, shape, goldens):
    """Test02VectorOperations (test/Test02VectorOperations.cpp:105-138): prints u.u == 0 for
    fp32/fp64 in 1/2/3-D with 10 colours (colours split the leading dimension, so the 3-D case
    has six empty pieces)."""
    n = int(np.prod(shape))
    per_slab = n // shape[0]
    pcs = [(lo * per_slab, (hi + 1) * per_slab - 1) for lo, hi in part.equal_partition(0, shape[0] - 1, 10)]
    pl = S.Planner(dtype)
    pl.add_sol_vector(np.zeros(n, dtype), pcs)
    pl.add_rhs_vector(np.zeros(n, dtype), pcs)
    pl.allocate_workspace(1)
    U, V, W = 0, 1, 2
    pl.get_vector(U, 0)[:] = 1.5
    pl.get_vector(V, 0)[:] = 2.7
    pl.copy(W, U)
    pl.axpy(W, dtype(1.0), V)
    pl.xpay(V, dtype(-1.0), U)
    pl.axpy(U, dtype(-0.5), V)
    pl.axpy(U, dtype(-0.5), W)
    assert f"{float(pl.dot(U, U)):g}" == goldens["test02_stdout"][0]


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_scalar_chain_test01(dtype, goldens):
    """Test01ScalarOperations: (2+10)/(2+2) - 2 == 1."""
    two, ten = dtype(2), dtype(10)
    op = lambda name, a, b=0.0: oracle.scalar_op(name, a, b, dtype)
    v = op("sub", op("div", op("add", two, ten), op("add", two, two)), two)
    assert v == goldens["test01_scalar_chain"]["value"]
    assert op("rsqrt", dtype(4)) == 0.5 and op("sqrt", dtype(9)) == 3 and op("neg", two) == -2


def test_get_alpha_rule():
    """src/LegionUtilities.cpp:72-97, operation order included."""
    f = [3.0, 7.0, 1.1, 0.3]
    assert oracle.get_alpha([], np.float64) == 1.0
    assert oracle.get_alpha(f[:1], np.float64) == 3.0
    assert oracle.get_alpha(f[:2], np.float64) == 3.0 / 7.0
    assert oracle.get_alpha(f[:3], np.float64) == (3.0 * 7.0) / 1.1
    assert oracle.get_alpha(f, np.float64) == (3.0 * 7.0) / (1.1 * 0.3)
    assert oracle.get_alpha(f[:3], np.float32) == (np.float32(3) * np.float32(7)) / np.float32(1.1)
    with pytest.raises(ValueError):
        oracle.get_alpha([1.0] * 5, np.float64)


def test_stencil_generator_equals_closed_form_laplacian():
    """The generic stencil fill on a 1-D grid writes what the closed-form Laplacian fill tasks
    write (src/ExampleSystems.cpp:311-321,382-389,448-466)."""
    for n in (2, 3, 20, 101):
        rects, row, col, val = oracle.laplacian_1d(n)
        offs, cfs = oracle.benchmark_stencil_offsets(1)
        r2, row2, col2, val2 = oracle.linearized_stencil((n,), offs, cfs, coo=True)
        assert np.array_equal(rects, r2) and np.array_equal(row, row2)
        assert np.array_equal(col, col2) and np.array_equal(val, val2)


@pytest.mark.parametrize("flag,extent", [(2, (7, 5)), (3, (4, 5, 6)), (4, (5, 4, 6)), (4, (2, 2, 2)), (4, (1, 3, 1))])
def test_stencil_matrices_against_scipy(flag, extent):
    """2-D/3-D stencils are not pinned by any reference golden: cross-check the generator and
    both matvecs with an independent scipy construction (Kronecker sums / explicit loops)."""
    offs, cfs = oracle.benchmark_stencil_offsets(flag)
    rects, row, col, val = oracle.linearized_stencil(extent, offs, cfs, coo=True)
    n = int(np.prod(extent))
    assert val.size == oracle.stencil_size(extent, offs)
    # independent dense construction
    A = np.zeros((n, n))
    grid = np.arange(n).reshape(extent)
    for o, c in zip(offs, cfs):
        for idx in np.ndindex(*extent):
            j = tuple(i + d for i, d in zip(idx, o))
            if all(0 <= jj < e for jj, e in zip(j, extent)):
                A[grid[idx], grid[j]] += c
    M = sp.coo_matrix((val, (row, col)), shape=(n, n)).toarray()
    assert np.array_equal(M, A)
    indptr = oracle.rects_to_indptr(rects)
    assert np.array_equal(indptr, sp.csr_matrix(A).indptr) or flag == 4  # explicit zeros never occur
    assert np.array_equal(rects[:, 1] + 1, indptr[1:])
    # column ids ascend within every row (offsets sorted lexicographically)
    for r in range(n):
        seg = col[rects[r, 0]:rects[r, 1] + 1]
        assert np.all(np.diff(seg) > 0)
    rng = np.random.default_rng(flag)
    x = rng.standard_normal(n)
    y_csr = np.zeros(n)
    oracle.csr_matvec(rects, 0, col, val, x, 0, n - 1, y_csr)
    y_lit = np.zeros(n)
    oracle.csr_matvec(rects, 0, col, val, x, 0, n - 1, y_lit, literal=True)
    y_coo = np.zeros(n)
    oracle.coo_matvec(row, col, val, x, 0, n - 1, y_coo, 0, n - 1)
    assert np.array_equal(y_csr, y_lit)  # O(nnz) restatement == the literal quadratic loop
    assert np.array_equal(y_csr, y_coo)  # same products in the same order
    np.testing.assert_allclose(y_csr, A @ x, rtol=1e-13, atol=1e-13)


def test_matvec_domain_guards():
    """Entries whose column (or row) lies outside the instance are skipped
    (src/COOMatrixTasks.cpp:71, src/CSRMatrixTasks.cpp:88)."""
    rects, row, col, val = oracle.laplacian_1d(10)
    x = np.arange(10, dtype=np.float64) + 1
    y = np.zeros(10)
    oracle.csr_matvec(rects, 0, col, val, np.ascontiguousarray(x[2:8]), 2, 7, y)
    A = sp.coo_matrix((val, (row, col)), shape=(10, 10)).toarray()
    A[:, :2] = 0
    A[:, 8:] = 0
    np.testing.assert_array_equal(y, A @ x)
    y2 = np.zeros(4)
    oracle.coo_matvec(row, col, val, x, 0, 9, y2, 3, 6)
    full = sp.coo_matrix((val, (row, col)), shape=(10, 10)).toarray() @ x
    np.testing.assert_array_equal(y2, full[3:7])


def test_task_ids():
    """src/TaskBaseClasses.hpp: block size 3*27*64 = 5184, origin 500000 + 1 meta id."""
    AXPY, CSR_MATVEC = 12, 18  # TaskIDs.hpp:17-52 (enum order)
    assert oracle.task_id_t(0, 0) == 500001
    assert oracle.task_id_tdi(1, 0, 1, 0) == 500001 + 5184
    # AxpyTask<double, 1, long long>: entry 1, dim 1, index type 2
    assert oracle.task_id_tdi(AXPY, 1, 1, 2) == 500001 + 5184 * AXPY + 3 * 3 * 2 + 0 + 1
    # CSRMatvecTask<double,1,1,1,ll,ll,ll>
    assert oracle.task_id_tdddiii(CSR_MATVEC, 1, 1, 1, 1, 2, 2, 2) == \
        500001 + 5184 * CSR_MATVEC + (16 + 4 + 1) * 3 * 2 + 1
    # all ids of one block are distinct and stay inside the block
    ids = {oracle.task_id_tdddiii(0, t, a, b, c, i, j, k)
           for t in range(2) for a in (1, 2, 3) for b in (1, 2, 3) for c in (1, 2, 3)
           for i in range(3) for j in range(3) for k in range(3)}
    assert len(ids) == 2 * 27 * 27 and max(ids) < 500001 + 5184


def test_equal_partition_rule():
    assert part.equal_partition(0, 19, 4) == [(0, 4), (5, 9), (10, 14), (15, 19)]
    assert part.equal_partition(0, 3, 10) == [(0, -1)] * 6 + [(0, 0), (1, 1), (2, 2), (3, 3)]
    sizes = [hi - lo + 1 for lo, hi in part.equal_partition(0, 9, 4)]
    assert sum(sizes) == 10 and max(sizes) - min(sizes) <= 1
