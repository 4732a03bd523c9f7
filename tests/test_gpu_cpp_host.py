"""The C++ host layer (host_cpp/: the reference's caller classes and its BenchmarkStencil program above
the C ABI, no Python or torch underneath; built by __graft_entry__.build()).  The program is run as
a child process and what it prints is compared with the reference's golden CG output
(test_all.py:130-133) and with the CPU oracle's solvers on the same systems."""
import os
import subprocess

import numpy as np
import pytest

from oracle import solvers as S

import host_cases as hc

pytestmark = pytest.mark.gpu

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BINARY = os.path.join(REPO, "host_cpp", "BenchmarkStencil")


def run(*flags):
    if not os.path.exists(BINARY):
        pytest.fail("host_cpp/BenchmarkStencil is not built: run __graft_entry__.build()")
    args = [str(f) for f in flags]
    print("case BenchmarkStencil " + " ".join(args), flush=True)
    res = subprocess.run([BINARY, *args], capture_output=True, text=True, timeout=200)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = res.stdout.splitlines()
    assert lines[0] == "Performed 5 warmup traces." and lines[1].startswith("Performed ")
    assert lines[2].startswith("Achieved ") and lines[2].endswith("ms per iteration.")
    values = [l for l in lines[3:] if not l[:1].isalpha()]
    sums = {l.split()[0]: float(l.split()[1]) for l in lines[3:] if l.startswith("solution_")}
    return [float(v) for v in values], sums, lines[3:]


@pytest.mark.parametrize("coo", [0, 1])
def test_reference_golden_cg_output(goldens, krylov_golden, coo):
    """Test05COOSolveCG / Test06CSRSolveCG: 1-D Laplacian N = 100, 4 pieces, 10 steps."""
    hist, _, raw = run("-dim", 1, "-nx", 100, "-vp", 4, "-it", 10, "-systems", 1, "-history", 1, "-coo", coo)
    assert sorted(raw[:11]) == sorted(goldens["expected_cg_output_sorted_set"])
    np.testing.assert_allclose(hist, krylov_golden["cases"][0]["residual_norm_squared"], rtol=1e-10)


@pytest.mark.parametrize("f32", [0, 1])
@pytest.mark.parametrize("solver", [1, 2, 3])
@pytest.mark.parametrize("coo", [0, 1])
def test_solvers_follow_the_oracle(solver, coo, f32):
    """27-point stencil 10 x 9 x 8 in 3 pieces, two systems as in the reference's program."""
    dtype = np.float32 if f32 else np.float64
    steps = {1: 8, 2: 6, 3: 6}[solver]
    rects, row, col, val, n = hc.stencil_problem(4, (10, 9, 8), dtype)
    opl = hc.oracle_planner("coo" if coo else "csr", rects, row, col, val, n, 3, dtype, spaces=2)
    so = {1: S.CGSolver, 2: S.BiCGStabSolver}[solver](opl) if solver < 3 else S.GMRESSolver(opl, 4)
    for _ in range(steps):
        so.step()
    hist, sums, _ = run("-dim", 4, "-nx", 10, "-ny", 9, "-nz", 8, "-vp", 3, "-solver", solver, "-it", steps,
                        "-restart", 4, "-history", 1, "-coo", coo, "-f32", f32)
    rtol = 2e-3 if f32 else 1e-9
    if solver == 1:
        np.testing.assert_allclose(hist, [float(v) for v in so.residual_norm_squared], rtol=rtol)
    elif solver == 2:
        np.testing.assert_allclose(hist, [float(v) for v in so.rho], rtol=rtol)
    x = opl.get_vector(0, 0).astype(np.float64)
    np.testing.assert_allclose(sums["solution_sum_of_squares"], float(np.sum(x * x)), rtol=rtol * 10)
    assert abs(sums["solution_sum"] - float(np.sum(x))) <= rtol * 10 * float(np.sum(np.abs(x)))


def test_compressed_structure_prints_the_same():
    """-compress 0 (products on the int64 columns and rects) and the 32-bit copy of the structure
    print the same history to the last digit (the sliced-pattern copy off in both); with it on, the
    history agrees up to the order of the sums."""
    flags = ("-dim", 3, "-nx", 20, "-ny", 17, "-nz", 33, "-vp", 2, "-it", 9, "-systems", 1, "-history", 1)
    c32 = run(*flags, "-sliced", 0)
    assert c32[2] == run(*flags, "-compress", 0, "-sliced", 0)[2]
    np.testing.assert_allclose(run(*flags)[0], c32[0], rtol=1e-11)


@pytest.mark.parametrize("solver", [1, 2, 3])
def test_held_back_updates_print_the_same_solution(solver):
    """-fuse 0 (every leaf launched as asked for) against the default (updates held back and fused
    with the xpay / dot that follows): the same history up to the summation order of the dots."""
    flags = ("-dim", 4, "-nx", 11, "-ny", 12, "-nz", 13, "-vp", 3, "-solver", solver, "-it", 7, "-restart", 4,
             "-history", 1)
    hist_f, sums_f, _ = run(*flags)
    hist_p, sums_p, _ = run(*flags, "-fuse", 0)
    np.testing.assert_allclose(hist_f, hist_p, rtol=1e-11)
    np.testing.assert_allclose(sums_f["solution_sum_of_squares"], sums_p["solution_sum_of_squares"], rtol=1e-10)


@pytest.mark.parametrize("coo,f32,chunk", [(0, 0, 1), (0, 0, 5000), (1, 0, 1), (0, 1, 1)])
def test_transposed_product(coo, f32, chunk):
    """-adjoint: x after 4 CG steps times the transpose (built in one run of entries, or in runs of
    5000) against x times the matrix, which is symmetric, and <A x, b> = <x, A^T b>."""
    _, _, raw = run("-dim", 4, "-nx", 12, "-ny", 11, "-nz", 10, "-vp", 3, "-it", 4, "-systems", 1, "-coo", coo,
                    "-f32", f32, "-adjoint", chunk)
    diff, _, scale = [l for l in raw if l.startswith("transposed_product_max_difference")][0].split()[1:]
    lhs, rhs = (float(v) for v in [l for l in raw if l.startswith("adjoint")][0].split()[1:])
    tol = 1e-5 if f32 else 1e-13
    assert float(scale) > 0.5 and float(diff) <= tol * float(scale)
    assert abs(lhs - rhs) <= 100 * tol * abs(lhs)


@pytest.mark.parametrize("solver", [1, 2])
@pytest.mark.parametrize("f32", [0, 1])
def test_binned_coo_follows_the_oracle(solver, f32):
    """-binned 1: the COO matrix multiplied through the binned copy (lsk_coo_binned_*: blocks of 64
    columns, tiles of 128 rows here, so the 720 rows make a dozen of each), in 3 pieces, two systems."""
    dtype = np.float32 if f32 else np.float64
    steps = {1: 8, 2: 6}[solver]
    rects, row, col, val, n = hc.stencil_problem(4, (10, 9, 8), dtype)
    opl = hc.oracle_planner("coo", rects, row, col, val, n, 3, dtype, spaces=2)
    so = {1: S.CGSolver, 2: S.BiCGStabSolver}[solver](opl)
    for _ in range(steps):
        so.step()
    hist, sums, _ = run("-dim", 4, "-nx", 10, "-ny", 9, "-nz", 8, "-vp", 3, "-solver", solver, "-it", steps,
                        "-history", 1, "-coo", 1, "-f32", f32, "-binned", 1, "-bincols", 64, "-binrows", 128)
    rtol = 2e-3 if f32 else 1e-9
    want = so.residual_norm_squared if solver == 1 else so.rho
    np.testing.assert_allclose(hist, [float(v) for v in want], rtol=rtol)
    x = opl.get_vector(0, 0).astype(np.float64)
    np.testing.assert_allclose(sums["solution_sum_of_squares"], float(np.sum(x * x)), rtol=rtol * 10)


@pytest.mark.parametrize("vp", [1, 3])
def test_classical_gram_schmidt_option_follows_the_oracle(vp):
    """-cgs 1: GMRES with every inner product of a step taken from the same vector
    (SquarePlanner::multi_dot / multi_axpy) against the oracle's restatement of that variant."""
    rects, row, col, val, n = hc.stencil_problem(4, (10, 9, 8), np.float64)
    opl = hc.oracle_planner("csr", rects, row, col, val, n, vp, np.float64, spaces=2)
    so = S.GMRESSolver(opl, 5, "classical")
    for _ in range(6):
        so.step()
    _, sums, _ = run("-dim", 4, "-nx", 10, "-ny", 9, "-nz", 8, "-vp", vp, "-solver", 3, "-it", 6, "-restart", 5,
                     "-history", 1, "-cgs", 1)
    x = opl.get_vector(0, 0).astype(np.float64)
    np.testing.assert_allclose(sums["solution_sum_of_squares"], float(np.sum(x * x)), rtol=1e-8)
    assert abs(sums["solution_sum"] - float(np.sum(x))) <= 1e-8 * float(np.sum(np.abs(x)))


@pytest.mark.parametrize("solver", [1, 2, 3])
def test_two_systems_in_one_pass_print_the_same(solver):
    """The program's two systems on one matrix, multiplied in one pass over its values (the default)
    and one after the other (-pair 0): the same history and solution sums to the last digit."""
    for grid, vp in (((13, 11, 17), 2), ((40, 16, 24), 1)):
        flags = ("-dim", 4, "-nx", grid[0], "-ny", grid[1], "-nz", grid[2], "-vp", vp, "-solver", solver, "-it", 7,
                 "-restart", 4, "-history", 1)
        paired, separate = run(*flags)[2], run(*flags, "-pair", 0)[2]
        count = lambda lines: int([l for l in lines if l.startswith("paired_products")][0].split()[1])
        assert count(separate) == 0
        assert [l for l in paired if not l.startswith("paired_products")] == \
               [l for l in separate if not l.startswith("paired_products")]
        if vp == 1:
            assert count(paired) > 0   # (a piece with a block that fits no common list makes the library decline the pair)


@pytest.mark.parametrize("solver", [1, 2, 3])
def test_scalar_coefficients_print_the_same(solver):
    """-uniform 1 (the library's option sliced.uniform: a constant-coefficient stencil's values as scalars,
    no value array read): the same history and solution sums to the last digit as the default."""
    for dim, grid, vp in ((4, (3, 4, 512), 2), (3, (9, 11, 13), 1), (2, (37, 41, 1), 1)):
        flags = ("-dim", dim, "-nx", grid[0], "-ny", grid[1], "-nz", grid[2], "-vp", vp, "-solver", solver, "-it", 7,
                 "-restart", 4, "-history", 1)
        assert run(*flags, "-uniform", 1)[2] == run(*flags)[2]


@pytest.mark.parametrize("solver,pt", [(1, 1), (1, 3), (2, 2), (3, 1)])
@pytest.mark.parametrize("coo", [0, 1])
def test_traces_recorded_as_hip_graphs_print_the_same(solver, pt, coo):
    """-graph 1: the first trace of -pt steps runs as it is, the next is recorded as a HIP graph
    (LegionSolversHost::Trace over lsk_stream_begin_capture / lsk_trace_replay) and every further trace is one
    replay, where the reference replays a Legion trace (test/BenchmarkStencil.cpp:219-222): the same last
    scalars and the same solution as the steps launched one by one, up to the order of the sums where a
    fused leaf would have reached across the end of a trace."""
    flags = ("-dim", 4, "-nx", 11, "-ny", 12, "-nz", 13, "-vp", 3, "-solver", solver, "-it", 7 * pt, "-pt", pt,
             "-restart", 4, "-history", 1, "-coo", coo)
    hist_e, sums_e, _ = run(*flags)
    hist_g, sums_g, raw = run(*flags, "-graph", 1)
    nodes = int([l for l in raw if l.startswith("trace_nodes")][0].split()[1])
    assert nodes >= 4 * pt   # a CG step is at least a product, two updates and a fold
    if solver in (1, 2):
        # the history holds the start, the first trace's steps and the LAST replay's steps; the entry the
        # recording started from is where every replay leaves its newest value for the next one
        assert len(hist_g) == 1 + 2 * pt and len(hist_e) == 1 + 7 * pt
        np.testing.assert_allclose(hist_g[:pt], hist_e[:pt], rtol=1e-12)
        np.testing.assert_allclose(hist_g[1 + pt:], hist_e[-pt:], rtol=1e-9)
        assert hist_g[pt] == hist_g[-1]
    np.testing.assert_allclose(sums_g["solution_sum_of_squares"], sums_e["solution_sum_of_squares"], rtol=1e-9)
    assert abs(sums_g["solution_sum"] - sums_e["solution_sum"]) <= 1e-9 * abs(sums_e["solution_sum"]) + 1e-12
