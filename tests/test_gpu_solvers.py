"""End-to-end GPU tests: the reference's test programs (Test01, Test02, Test05, Test06) and the
BiCGStab / GMRES sequences run through the host mirror on the HIP backend and compared with the
reference's goldens and the CPU oracle (fp64 residual histories within 1e-10 relative, the bar
the north star states); plus size-independent properties of the SpMV at the headline size."""
import numpy as np
import pytest
import torch

import oracle
from oracle import solvers as S

import host_cases as hc

from legionsolvers_amd import (BiCGStabSolver, CGSolver, GMRESSolver, IndexPartition, IndexSpace,
                               PartitionedVector, Scalar, SquarePlanner)
from legionsolvers_amd import stencil_generator as sg

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("kind", ["coo", "csr"])
@pytest.mark.parametrize("pieces", [1, 4])
def test_cg_reference_golden(gpu_ctx, kind, pieces, goldens, krylov_golden):
    """Test05COOSolveCG / Test06CSRSolveCG: N = 100, 10 CG steps, compared as the reference's
    driver does (sorted printed values, test_all.py:130-133) and, in order, with the outputs of
    the reference's scripts/krylov.py."""
    print(f"case cg golden {kind} pieces={pieces}", flush=True)
    prob = goldens["cg_problem"]
    ip = IndexPartition.equal(IndexSpace(0, prob["n"] - 1), pieces)
    make = sg.csr_negative_laplacian_1d if kind == "csr" else sg.coo_negative_laplacian_1d
    matrix = make(gpu_ctx, prob["n"], ip)
    rhs = PartitionedVector(gpu_ctx, "rhs", ip)
    rhs.constant_fill(1.0)
    sol = PartitionedVector(gpu_ctx, "sol", ip)
    sol.zero_fill()
    planner = SquarePlanner(gpu_ctx)
    planner.add_sol_vector(sol)
    planner.add_rhs_vector(rhs)
    planner.add_row_partitioned_matrix(matrix, 0, 0)
    solver = CGSolver(planner)
    for _ in range(prob["iterations"]):
        solver.step()
    hist = [s.get_value() for s in solver.residual_norm_squared]
    golden = krylov_golden["cases"][0]
    np.testing.assert_allclose(hist, golden["residual_norm_squared"], rtol=1e-10)
    assert sorted(f"{v:g}" for v in hist) == sorted(goldens["expected_cg_output_sorted_set"])
    np.testing.assert_allclose(sol.to_host().numpy(), golden["x_final"], rtol=1e-10)


def test_vector_and_scalar_programs(gpu_ctx, goldens):
    """Test02VectorOperations (u.u prints 0, fp32/fp64 x 1/2/3-D, 10 colours, some of them
    empty) and Test01ScalarOperations ((2+10)/(2+2)-2 == 1)."""
    for dtype in (torch.float32, torch.float64):
        for space in (IndexSpace(0, 99), IndexSpace((0, 0), (9, 9)), IndexSpace((0, 0, 0), (3, 4, 4))):
            print(f"case test02 {dtype} {space}", flush=True)
            ip = IndexPartition.equal(space, 10)
            u, v, w = (PartitionedVector(gpu_ctx, n, ip, dtype) for n in "uvw")
            u.constant_fill(1.5)
            v.constant_fill(2.7)
            w.assign(u)
            w.axpy(1.0, v)
            v.xpay(-1.0, u)
            u.axpy(-0.5, v)
            u.axpy(-0.5, w)
            assert f"{u.dot(u).get_value():g}" == goldens["test02_stdout"][0]
        two, ten = Scalar(gpu_ctx, 2.0, dtype), Scalar(gpu_ctx, 10.0, dtype)
        assert ((two + ten) / (two + two) - two).get_value() == goldens["test01_scalar_chain"]["value"]


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("solver_name", ["cg", "bicgstab", "gmres"])
@pytest.mark.parametrize("kind", ["csr", "coo"])
def test_solvers_match_oracle(gpu_ctx, dtype, solver_name, kind):
    """CG / BiCGStab / GMRES on a 3-D 27-point system with a random right-hand side, 4 pieces,
    two coupled spaces for CG (the BenchmarkStencil set-up): solution and scalar histories
    against the oracle's restatement of the reference headers."""
    print(f"case solver {solver_name} {kind} {np.dtype(dtype).name}", flush=True)
    rects, row, col, val, n = hc.stencil_problem(4, (8, 9, 10), dtype)
    rhs = np.random.default_rng(31).standard_normal(n)
    spaces = 2 if solver_name == "cg" else 1
    opl = hc.oracle_planner(kind, rects, row, col, val, n, 4, dtype, rhs, spaces)
    mpl = hc.mirror_planner(gpu_ctx, kind, rects, row, col, val, n, 4, dtype, rhs, spaces)
    if solver_name == "cg":
        so, sm, steps = S.CGSolver(opl), CGSolver(mpl), 8
    elif solver_name == "bicgstab":
        so, sm, steps = S.BiCGStabSolver(opl), BiCGStabSolver(mpl), 5
    else:
        so, sm, steps = S.GMRESSolver(opl, 5), GMRESSolver(mpl, 5), 2
    for _ in range(steps):
        so.step()
        sm.step()
    rtol = 1e-10 if dtype == np.float64 else 2e-3
    if solver_name == "cg":
        np.testing.assert_allclose([s.get_value() for s in sm.residual_norm_squared],
                                   [float(v) for v in so.residual_norm_squared], rtol=rtol)
    if solver_name == "bicgstab":
        np.testing.assert_allclose([s.get_value() for s in sm.rho], [float(v) for v in so.rho], rtol=rtol)
    want = opl.get_vector(0, spaces - 1).astype(np.float64)
    got = mpl.get_vector(0, spaces - 1).to_host().numpy().astype(np.float64)
    scale = np.abs(want).max()
    assert np.abs(got - want).max() <= rtol * 10 * scale


@pytest.mark.parametrize("n1", [512], ids=["512cubed"])
def test_spmv_properties_at_headline_size(gpu_ctx, monkeypatch, n1):
    """BenchmarkStencil's headline matrix (3-D 27-point, 512^3, fp64: 134M rows, 3.6G entries,
    58 GB of col+val) is far beyond what the CPU oracle can check; these properties are
    size-independent: interior row sums are 0 (88 - 6*6 - 12*3 - 8*2 = 0), boundary row sums
    are positive, A is symmetric (x.Ay == y.Ax), A is linear, and the entry count is
    (3*512-2)^3."""
    from conftest import require_device_memory
    require_device_memory(190, f"27-point fp64 CSR SpMV properties, two-vector / uniform-coefficient / updated-values "
                               f"sliced products at {n1}^3 (3 609 741 304 entries)")
    n = n1 ** 3
    print(f"case spmv properties {n1}^3", flush=True)
    dim, offs, cfs = sg.benchmark_stencil(4)
    ip = IndexPartition.equal(IndexSpace(0, n - 1), 1)
    A = sg.create_linearized_csr_stencil_matrix(gpu_ctx, (n1, n1, n1), offs, cfs, ip)
    assert A.nnz == (3 * n1 - 2) ** 3
    be = gpu_ctx.backend
    kp = A.create_kernel_partition_from_range_partition(ip)
    assert kp.intervals == [(0, A.nnz - 1)]
    gp = A.create_domain_partition_from_kernel_partition(IndexSpace(0, n - 1), kp)
    assert gp.intervals == [(0, n - 1)]

    def matvec(x):
        y = PartitionedVector(gpu_ctx, "y", ip)
        A.matvec(y, x.owned, 0, n - 1, kp, 0)
        return y

    ones = PartitionedVector(gpu_ctx, "ones", ip)
    ones.constant_fill(1.0)
    rowsum = matvec(ones).owned.view(n1, n1, n1)
    interior = rowsum[1:-1, 1:-1, 1:-1]
    assert float(interior.abs().max()) <= 1e-14
    assert float(rowsum[0].min()) > 0.1 and float(rowsum[:, 0].min()) > 0.1 and float(rowsum[:, :, -1].min()) > 0.1
    del rowsum, interior
    g = torch.Generator(device="cuda").manual_seed(5)
    x = PartitionedVector(gpu_ctx, "x", ip)
    y = PartitionedVector(gpu_ctx, "yv", ip)
    x.owned.copy_(torch.randn(n, generator=g, device="cuda", dtype=torch.float64))
    y.owned.copy_(torch.randn(n, generator=g, device="cuda", dtype=torch.float64))
    Ax, Ay = matvec(x), matvec(y)
    xAy, yAx = x.dot(Ay).get_value(), y.dot(Ax).get_value()
    norm = (x.dot(x).get_value() * Ay.dot(Ay).get_value()) ** 0.5
    assert abs(xAy - yAx) <= 1e-12 * norm
    # linearity: A(2x - 3y) == 2Ax - 3Ay
    z = PartitionedVector(gpu_ctx, "z", ip)
    z.assign(x)
    z.scal(2.0)
    z.axpy(-3.0, y)
    Az = matvec(z)
    Az.axpy(-2.0, Ax)
    Az.axpy(3.0, Ay)
    assert Az.dot(Az).get_value() ** 0.5 <= 1e-12 * (4 * Ax.dot(Ax).get_value() + 9 * Ay.dot(Ay).get_value()) ** 0.5
    # the two set-up-time layouts the planner puts this matrix in, against the product above (which
    # read the int64 arrays): the 32-bit structure bit for bit, the sliced-pattern copy (27 patterns,
    # the two end rows of every grid line masked, rows summed in entry order) to rounding
    del Az, Ay, z, y
    monkeypatch.setenv("LSK_TUNE_REDRAWS", "0")
    assert A.compress(0, n - 1, kp, ip)
    assert torch.equal(matvec(x).owned, Ax.owned)
    assert A.slice_structure(0, n - 1, kp, ip)
    s = A.sliced[0]
    # a block is one grid line (512 rows): its first and its last wave hold a line end each
    assert (s.patterns, s.mixed_blocks, s.masked_waves) == (27, 0, 2 * s.blocks) and s.blocks * s.block_rows == n
    # padding: the two end rows of every grid line have 2/3 of the line's row length (27, 18 or 12)
    assert s.padded_entries == A.nnz + 2 * (9 * (n1 - 2) ** 2 + 6 * 4 * (n1 - 2) + 4 * 4)
    As = matvec(x)
    diff = float((As.owned - Ax.owned).abs().max())
    assert 0.0 < diff <= 1e-13 * float(Ax.owned.abs().max())
    # What the benchmark times beyond the plain product, at this size, bit for bit against it:
    # (a) two right-hand sides in one pass over the values (lsk_csr_sliced_spmv2_*), with the dot shares
    print(f"case two-vector sliced product {n1}^3", flush=True)
    from legionsolvers_amd import _lib
    del Ax
    x2 = PartitionedVector(gpu_ctx, "x2", ip)
    x2.owned.copy_(torch.randn(n, generator=g, device="cuda", dtype=torch.float64))
    As2 = matvec(x2)
    ya, yb = (torch.full((n,), 9.0, dtype=torch.float64, device="cuda") for _ in range(2))
    assert be.csr_sliced_spmv2(s, n, [x.owned, x2.owned], [ya, yb], 0) == 0
    assert torch.equal(ya, As.owned) and torch.equal(yb, As2.owned)
    slots = be.csr_sliced_dot_slots(s, n)
    single = [torch.empty(slots, dtype=torch.float64, device="cuda") for _ in range(2)]
    pair = [torch.empty(slots, dtype=torch.float64, device="cuda") for _ in range(2)]
    ya.fill_(9.0), yb.fill_(9.0)
    for xv, out in ((x, single[0]), (x2, single[1])):
        scratch = torch.empty(n, dtype=torch.float64, device="cuda")
        assert be.csr_sliced_spmv_dot(s, n, xv.owned, scratch, 0, xv.owned, out) == slots
        assert torch.equal(scratch, As.owned if xv is x else As2.owned)
        del scratch
    assert be.csr_sliced_spmv2(s, n, [x.owned, x2.owned], [ya, yb], 0, [x.owned, x2.owned], pair) == slots
    assert torch.equal(ya, As.owned) and torch.equal(yb, As2.owned)
    assert torch.equal(pair[0], single[0]) and torch.equal(pair[1], single[1])
    del yb, As2, x2, single, pair
    # (b) the values as scalars (option sliced.uniform): the matrix is a constant-coefficient stencil
    print(f"case uniform-coefficient sliced product {n1}^3", flush=True)
    assert s.uniform_values
    _lib.call("lsk_set_option", b"sliced.uniform", 1)
    try:
        ya.fill_(9.0)
        be.csr_sliced_spmv(s, 0, n, x.owned, ya, 0)
        assert torch.equal(ya, As.owned)
        # (c) values rewritten in place, then update_values: equal to a fresh copy of the new values; the
        # coefficients are no longer uniform, and the option falls back to reading them
        print(f"case update_values {n1}^3", flush=True)
        chunk = 1 << 28
        for a in range(0, A.nnz, chunk):
            piece = A.val[a:a + chunk]
            piece.mul_(torch.rand(piece.numel(), generator=g, device="cuda", dtype=torch.float64) + 0.5)
            del piece
        A.values_changed()
        assert not s.uniform_values
        be.csr_sliced_spmv(s, 0, n, x.owned, ya, 0)
    finally:
        _lib.call("lsk_set_option", b"sliced.uniform", 0)
    fresh = be.csr_sliced_create(n, A.nnz, A.rects, 0, A.col, A.val, 0, 0, n - 1)
    assert fresh is not None
    try:
        y_fresh = torch.full((n,), 9.0, dtype=torch.float64, device="cuda")
        be.csr_sliced_spmv(fresh, 0, n, x.owned, y_fresh, 0)
        assert torch.equal(ya, y_fresh) and not torch.equal(ya, As.owned)
        be.csr_sliced_spmv(s, 0, n, x.owned, ya, 0)          # and without the option
        assert torch.equal(ya, y_fresh)
    finally:
        fresh.close()


def test_a_piece_of_more_than_2_to_the_32_entries(gpu_ctx, monkeypatch):
    """One piece with more entries than 32-bit offsets can address (27-point stencil on 544^3: 4 330 747 000
    entries, 161 M rows; the reference's int64 kernel coordinates exist for exactly this).  The products on
    the arrays as handed over must be right (row sums, symmetry), the 32-bit copy of the structure must be
    declined, not wrapped around, and whatever the planner picks instead must agree with the plain product
    and run CG like the plain arrays do."""
    from conftest import require_device_memory
    n1 = 544
    n = n1 ** 3
    require_device_memory(200, f"27-point fp64 CSR products and CG on one piece of {(3 * n1 - 2) ** 3} entries "
                               f"(more than 2^32) at {n1}^3")
    print(f"case more than 2^32 entries {n1}^3", flush=True)
    monkeypatch.setenv("LSK_TUNE", "0")
    dim, offs, cfs = sg.benchmark_stencil(4)
    ip = IndexPartition.equal(IndexSpace(0, n - 1), 1)
    A = sg.create_linearized_csr_stencil_matrix(gpu_ctx, (n1, n1, n1), offs, cfs, ip)
    assert A.nnz == (3 * n1 - 2) ** 3 and A.nnz > 2 ** 32
    kp = A.create_kernel_partition_from_range_partition(ip)
    assert kp.intervals == [(0, A.nnz - 1)]
    # the last rect ends at the last entry: the generator's offsets did not wrap
    assert A.rects[-1].tolist()[1] == A.nnz - 1 and int(A.col[-1]) == n - 1

    def matvec(x):
        y = PartitionedVector(gpu_ctx, "y", ip)
        A.matvec(y, x.owned, 0, n - 1, kp, 0)
        return y

    ones = PartitionedVector(gpu_ctx, "ones", ip)
    ones.constant_fill(1.0)
    rowsum = matvec(ones).owned.view(n1, n1, n1)
    assert float(rowsum[1:-1, 1:-1, 1:-1].abs().max()) <= 1e-14
    assert float(rowsum[0].min()) > 0.1 and float(rowsum[-1].min()) > 0.1 and float(rowsum[:, :, -1].min()) > 0.1
    del rowsum
    g = torch.Generator(device="cuda").manual_seed(11)
    x = PartitionedVector(gpu_ctx, "x", ip)
    y = PartitionedVector(gpu_ctx, "yv", ip)
    x.owned.copy_(torch.randn(n, generator=g, device="cuda", dtype=torch.float64))
    y.owned.copy_(torch.randn(n, generator=g, device="cuda", dtype=torch.float64))
    Ax, Ay = matvec(x), matvec(y)
    xAy, yAx = x.dot(Ay).get_value(), y.dot(Ax).get_value()
    assert abs(xAy - yAx) <= 1e-12 * (x.dot(x).get_value() * Ay.dot(Ay).get_value()) ** 0.5
    # the fused product + dot on the same arrays: same y, the dot to rounding
    be = gpu_ctx.backend
    slots = be.csr_spmv_dot_slots(n, A.nnz, torch.float64)
    partials = torch.empty(slots, dtype=torch.float64, device="cuda")
    yd = torch.full((n,), 9.0, dtype=torch.float64, device="cuda")
    count = be.csr_spmv_dot(n, A.nnz, A.rects, 0, A.col, A.val, x.owned, 0, n - 1, yd, 0, y.owned, partials)
    assert torch.equal(yd, Ax.owned)
    assert abs(float(partials[:count].sum()) - yAx) <= 1e-12 * abs(yAx) + 1e-6
    del yd, partials, Ay, y
    # registered with a planner: 32-bit offsets cannot hold this piece (declined, not wrapped around); the
    # sliced-pattern copy counts entries in 64 bits and is made; its product equals the plain one to rounding
    from legionsolvers_amd import benchmark_stencil as bs
    _, planner, solver = bs.build_problem(gpu_ctx, 4, n1, n1, n1, 1, 1, torch.float64, num_systems=1, matrix=A)
    assert A.col32 is None and not A.rowoff
    assert A.sliced and A.sliced_window == (0, n - 1)
    s = A.sliced[0]
    assert s.patterns == 27 and s.padded_entries > 2 ** 32
    As = matvec(x)
    diff = float((As.owned - Ax.owned).abs().max())
    assert 0.0 < diff <= 1e-13 * float(Ax.owned.abs().max())
    del As, Ax, x, ones
    # CG, three steps: on the copy, then on the arrays as handed over
    histories = []
    for _ in range(3):
        solver.step()
    histories.append([r.get_value() for r in solver.residual_norm_squared])
    del planner, solver
    held, A.sliced_window = A.sliced_window, None
    monkeypatch.setenv("LSK_SLICED", "0")   # (registering the matrix again must not make the copy again)
    try:
        _, planner, solver = bs.build_problem(gpu_ctx, 4, n1, n1, n1, 1, 1, torch.float64, num_systems=1, matrix=A)
        assert A.sliced_window is None and A.col32 is None
        for _ in range(3):
            solver.step()
        histories.append([r.get_value() for r in solver.residual_norm_squared])
    finally:
        A.sliced_window = held
    assert len(histories[0]) == 4 and histories[0][0] == float(n)
    assert all(np.isfinite(h) and h > 0.0 for h in histories[0])
    np.testing.assert_allclose(histories[0], histories[1], rtol=1e-10)


def test_a_coo_piece_of_more_than_2_to_the_32_entries(gpu_ctx, monkeypatch):
    """The same for the COO form (row, col int64 and val: 104 GB): the scan kernel on the arrays as handed
    over against row sums and symmetry, then the planner's choice for that piece against it in CG."""
    from conftest import require_device_memory
    from legionsolvers_amd import benchmark_stencil as bs
    n1 = 544
    n = n1 ** 3
    require_device_memory(230, f"27-point fp64 COO products and CG on one piece of {(3 * n1 - 2) ** 3} entries "
                               f"(more than 2^32) at {n1}^3")
    print(f"case coo more than 2^32 entries {n1}^3", flush=True)
    monkeypatch.setenv("LSK_TUNE", "0")
    dim, offs, cfs = sg.benchmark_stencil(4)
    ip = IndexPartition.equal(IndexSpace(0, n - 1), 1)
    A = sg.create_linearized_coo_stencil_matrix(gpu_ctx, (n1, n1, n1), offs, cfs, ip, torch.float64)
    assert A.nnz == (3 * n1 - 2) ** 3 and A.nnz > 2 ** 32
    assert int(A.row[-1]) == n - 1 and int(A.col[-1]) == n - 1 and int(A.row[2 ** 32]) > 0
    kp = A.create_kernel_partition_from_range_partition(ip)
    assert kp.intervals == [(0, A.nnz - 1)]

    def matvec(x):
        y = PartitionedVector(gpu_ctx, "y", ip)
        y.zero_fill()
        A.matvec(y, x.owned, 0, n - 1, kp, 1)
        return y

    ones = PartitionedVector(gpu_ctx, "ones", ip)
    ones.constant_fill(1.0)
    rowsum = matvec(ones).owned.view(n1, n1, n1)
    assert float(rowsum[1:-1, 1:-1, 1:-1].abs().max()) <= 1e-14
    assert float(rowsum[0].min()) > 0.1 and float(rowsum[-1].min()) > 0.1 and float(rowsum[:, :, -1].min()) > 0.1
    del rowsum, ones
    g = torch.Generator(device="cuda").manual_seed(12)
    x = PartitionedVector(gpu_ctx, "x", ip)
    y = PartitionedVector(gpu_ctx, "yv", ip)
    x.owned.copy_(torch.randn(n, generator=g, device="cuda", dtype=torch.float64))
    y.owned.copy_(torch.randn(n, generator=g, device="cuda", dtype=torch.float64))
    Ax, Ay = matvec(x), matvec(y)
    xAy, yAx = x.dot(Ay).get_value(), y.dot(Ax).get_value()
    assert abs(xAy - yAx) <= 1e-12 * (x.dot(x).get_value() * Ay.dot(Ay).get_value()) ** 0.5
    del Ax, Ay, x, y
    histories = []
    for plain in (False, True):
        if plain:
            monkeypatch.setenv("LSK_SLICED", "0")
            monkeypatch.setenv("LSK_COMPRESS", "0")
            A.sliced_window = A.col32_window = None   # (what the first registration made stays unused)
        _, planner, solver = bs.build_problem(gpu_ctx, 4, n1, n1, n1, 1, 1, torch.float64, num_systems=1, matrix=A)
        if plain:
            assert A.sliced_window is None and A.col32_window is None and A.binned is None
        else:   # a row-sorted stencil: the sliced-pattern copy, entries counted in 64 bits
            assert A.sliced and A.sliced_window == (0, n - 1) and A.sliced[0].padded_entries > 2 ** 32
        for _ in range(3):
            solver.step()
        histories.append([r.get_value() for r in solver.residual_norm_squared])
        del planner, solver
    assert histories[0][0] == float(n) and all(np.isfinite(h) and h > 0.0 for h in histories[0])
    np.testing.assert_allclose(histories[0], histories[1], rtol=1e-10)


@pytest.mark.parametrize("sliced", ["0", "1"])
def test_cg_with_tuned_placement(gpu_ctx, monkeypatch, sliced):
    """The planner's set-up-time placement tuning and array redraws (normally only for pieces of
    millions of rows) forced on a small problem, for the product on the 32-bit structure and for the
    one on the sliced-pattern copy: same CG history as the oracle, whatever placement it picks."""
    from legionsolvers_amd import CSRMatrix
    monkeypatch.setattr(CSRMatrix, "TUNE_MIN_ROWS", 1)
    monkeypatch.setenv("LSK_TUNE", "2")   # (placement timing is the default; the redraws are opt-in)
    monkeypatch.setenv("LSK_SLICED", sliced)
    rects, row, col, val, n = hc.stencil_problem(4, (16, 16, 16), np.float64)
    rhs = np.random.default_rng(32).standard_normal(n)
    opl = hc.oracle_planner("csr", rects, row, col, val, n, 2, np.float64, rhs)
    mpl = hc.mirror_planner(gpu_ctx, "csr", rects, row, col, val, n, 2, np.float64, rhs)
    matrix = mpl.row_partitioned_matrices[0].matrix
    chosen = matrix.sliced_placement if sliced == "1" else matrix.placement
    assert bool(matrix.sliced) == (sliced == "1")
    assert sorted(chosen) == [0, 1] and all(v in (0, 2, 4, 16, 64, 256, 1024, 4096) for v in chosen.values())
    so, sm = S.CGSolver(opl), CGSolver(mpl)
    for _ in range(8):
        so.step()
        sm.step()
    np.testing.assert_allclose([s.get_value() for s in sm.residual_norm_squared],
                               [float(v) for v in so.residual_norm_squared], rtol=1e-10)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("solver_name", ["cg", "gmres"])
def test_compressed_structure_gives_the_same_history(gpu_ctx, monkeypatch, dtype, solver_name):
    """The planner's set-up-time 32-bit copy of the CSR structure (on by default) and the products
    straight on the reference's int64 columns and rects (LSK_COMPRESS=0) give bit-identical
    residual histories: same tiles, same order of additions.  (The sliced-pattern structure, which a
    stencil matrix would get on top, is switched off here.)"""
    monkeypatch.setenv("LSK_SLICED", "0")
    rects, row, col, val, n = hc.stencil_problem(4, (20, 12, 16), dtype)
    rhs = np.random.default_rng(33).standard_normal(n).astype(dtype)
    histories = []
    for compress in ("1", "0"):
        monkeypatch.setenv("LSK_COMPRESS", compress)
        mpl = hc.mirror_planner(gpu_ctx, "csr", rects, row, col, val, n, 3, dtype, rhs)
        matrix = mpl.row_partitioned_matrices[0].matrix
        assert (matrix.col32 is not None) == (compress == "1")
        assert (sorted(matrix.rowoff) == [0, 1, 2]) == (compress == "1")
        solver = CGSolver(mpl) if solver_name == "cg" else GMRESSolver(mpl, 5)
        for _ in range(7 if solver_name == "cg" else 2):
            solver.step()
        histories.append(mpl.get_vector(0, 0).to_host().clone())
        if solver_name == "cg":
            histories[-1] = (histories[-1], [s.get_value() for s in solver.residual_norm_squared])
    if solver_name == "cg":
        assert histories[0][1] == histories[1][1] and torch.equal(histories[0][0], histories[1][0])
    else:
        assert torch.equal(histories[0], histories[1])


def hold_device_memory_except(keep: int):
    """Tensors that hold all of the device's free memory except about `keep` bytes (what the driver reports as
    free is not what one allocation gets: asked again until the report is down to `keep`)."""
    import gc
    held = []
    gc.collect()   # (what earlier tests left for the collector holds device memory that an allocation in need would find)
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    for _ in range(24):
        free, _ = torch.cuda.mem_get_info()
        if free <= keep + (4 << 20):
            break
        want = free - keep
        try:
            held.append(torch.empty(want, dtype=torch.uint8, device="cuda"))
        except torch.cuda.OutOfMemoryError:
            try:
                held.append(torch.empty(max(want // 2, 1 << 20), dtype=torch.uint8, device="cuda"))
            except torch.cuda.OutOfMemoryError:
                break
    # torch's allocator keeps memory of its own that it has reserved and not handed out (parts of segments that
    # live tensors sit in), a gigabyte after a long test session: taken as well, or it serves what is asked next
    for size in (256 << 20, 64 << 20, 16 << 20, 4 << 20, 1 << 20):
        for _ in range(2048):
            reserved = torch.cuda.memory_reserved()
            try:
                t = torch.empty(size, dtype=torch.uint8, device="cuda")
            except torch.cuda.OutOfMemoryError:
                break
            if torch.cuda.memory_reserved() > reserved:   # that one came from the driver: back it goes
                del t
                torch.cuda.empty_cache()
                break
            held.append(t)
    return held


def test_a_set_up_time_copy_that_does_not_fit_is_left_out(gpu_ctx, monkeypatch):
    """The set-up-time copies are optional: when the device has no room for the sliced-pattern copy (here: all
    but 64 MiB of the free memory is held while the library makes it, and the copy needs 190 MB), the library
    reports the failed allocation, leaks nothing and leaves no error behind for later launches; the planner
    notes it, gives the matrix the 32-bit copy of its structure instead, and CG follows the oracle."""
    be = gpu_ctx.backend
    rects, row, col, val, n = hc.stencil_problem(4, (96, 96, 96), np.float64)
    rhs = np.random.default_rng(35).standard_normal(n)
    real = be.csr_sliced_create
    seen = {}

    def with_the_memory_held(*args, **kwargs):
        torch.cuda.synchronize()
        torch.cuda.empty_cache()
        seen["free_before"], _ = torch.cuda.mem_get_info()
        held = hold_device_memory_except(64 << 20)
        seen["free_while_held"], _ = torch.cuda.mem_get_info()
        try:
            return real(*args, **kwargs)
        finally:
            del held
            torch.cuda.empty_cache()
            seen["free_after"], _ = torch.cuda.mem_get_info()

    monkeypatch.setattr(be, "csr_sliced_create", with_the_memory_held)
    print("case sliced-pattern copy with the device's memory held", flush=True)
    mpl = hc.mirror_planner(gpu_ctx, "csr", rects, row, col, val, n, 1, np.float64, rhs)
    matrix = mpl.row_partitioned_matrices[0].matrix
    assert seen["free_while_held"] <= (150 << 20)    # (the copy needs 190 MB)
    assert not matrix.sliced and matrix.col32 is not None
    assert any("no room for the sliced-pattern copy" in note and "OutOfMemory" in note for note in matrix.setup_notes)
    assert seen["free_after"] >= seen["free_before"] - (8 << 20)        # what the failed create had allocated is back
    opl = hc.oracle_planner("csr", rects, row, col, val, n, 1, np.float64, rhs)
    so, sm = S.CGSolver(opl), CGSolver(mpl)
    for _ in range(5):
        so.step()
        sm.step()
    np.testing.assert_allclose([s.get_value() for s in sm.residual_norm_squared],
                               [float(v) for v in so.residual_norm_squared], rtol=1e-10)


def test_set_up_time_copies_are_given_up_when_a_vector_finds_no_room(gpu_ctx):
    """The copies only make products faster; a vector that cannot be allocated ends the solve.  With all but
    100 MB of the device held, a vector of 150 MB is asked for: the matrix gives up its sliced-pattern copy
    (190 MB), takes the 32-bit copy of its structure instead (99 MB), the vector is allocated, and CG goes on
    following the oracle -- unless a recorded step still refers to the copy, then the allocation fails."""
    be = gpu_ctx.backend
    rects, row, col, val, n = hc.stencil_problem(4, (96, 96, 96), np.float64)
    rhs = np.random.default_rng(36).standard_normal(n)
    opl = hc.oracle_planner("csr", rects, row, col, val, n, 1, np.float64, rhs)
    mpl = hc.mirror_planner(gpu_ctx, "csr", rects, row, col, val, n, 1, np.float64, rhs)
    matrix = mpl.row_partitioned_matrices[0].matrix
    assert matrix.sliced and matrix.col32 is None
    so, sm = S.CGSolver(opl), CGSolver(mpl)
    for _ in range(3):
        so.step()
        sm.step()
    recorded = sm.capture()
    recorded.replay()          # the fourth step
    so.step()
    print("case a vector allocated with the device's memory held", flush=True)
    held = hold_device_memory_except(100 << 20)
    try:
        assert torch.cuda.mem_get_info()[0] <= (120 << 20)
        with pytest.raises(torch.cuda.OutOfMemoryError):   # the recording keeps the copy
            be.empty(150 << 17, torch.float64)
        assert matrix.sliced
        del recorded
        held += hold_device_memory_except(100 << 20)       # (what the recording itself had held)
        vector = be.empty(150 << 17, torch.float64)        # 150 MiB
        assert vector.numel() == 150 << 17
        assert not matrix.sliced and matrix.col32 is not None
        assert any("given up" in note for note in matrix.setup_notes)
    finally:
        del held
        torch.cuda.empty_cache()
    for _ in range(3):
        so.step()
        sm.step()
    got = [s.get_value() for s in sm.residual_norm_squared]
    want = [float(v) for v in so.residual_norm_squared]
    assert len(got) == len(want) == 8
    want[3] = want[4]          # (the entry the recording started from holds what its replay left there)
    np.testing.assert_allclose(got, want, rtol=1e-10)


def test_blocks_held_in_front_of_the_workspace_go_before_any_copy(gpu_ctx):
    """While the planner holds memory in front of its workspace (to steer where the vectors land), a vector that
    finds no room must let that memory go, not make a matrix give up a set-up-time copy.  Forced here by a planner
    whose held memory is all of the device but 10 MB, against a workspace of 21 MB."""
    rects, row, col, val, n = hc.stencil_problem(4, (96, 96, 96), np.float64)
    rhs = np.random.default_rng(37).standard_normal(n)
    opl = hc.oracle_planner("csr", rects, row, col, val, n, 1, np.float64, rhs)
    mpl = hc.mirror_planner(gpu_ctx, "csr", rects, row, col, val, n, 1, np.float64, rhs)
    matrix = mpl.row_partitioned_matrices[0].matrix
    assert matrix.sliced
    asked = []

    def all_but_10_mb(workspace_bytes=0):
        asked.append(workspace_bytes)
        mpl.workspace_placement = {"blocks_in_front": 1, "product_ms_by_blocks_in_front": [2.0, 1.0]}
        return hold_device_memory_except(10 << 20)

    mpl._room_in_front_of_the_workspace = all_but_10_mb
    print("case workspace allocated behind memory that leaves it no room", flush=True)
    so, sm = S.CGSolver(opl), CGSolver(mpl)
    assert asked == [3 * n * 8]
    assert matrix.sliced and not any("given up" in note for note in matrix.setup_notes)
    assert mpl.workspace_placement["blocks_in_front"] == 0
    assert gpu_ctx.backend.optional_memory_stays is False
    assert torch.cuda.mem_get_info()[0] > (1 << 30)      # the held memory is back
    for _ in range(4):
        so.step()
        sm.step()
    np.testing.assert_allclose([s.get_value() for s in sm.residual_norm_squared],
                               [float(v) for v in so.residual_norm_squared], rtol=1e-10)


def test_a_coo_matrix_gives_up_its_binned_copy(gpu_ctx, monkeypatch):
    """The binned copy of an unstructured COO matrix is optional memory as well: asked to release it in the middle
    of a BiCGStab run (what an allocation that finds no room does), the matrix goes on scanning its entries and
    the solver goes on following the oracle."""
    from legionsolvers_amd import COOMatrix
    monkeypatch.setenv("LSK_COO_BLOCK", "1")
    monkeypatch.setenv("LSK_COO_BINNED", "2")
    monkeypatch.setattr(COOMatrix, "BLOCK_MIN_WINDOW", 0)
    monkeypatch.setattr(COOMatrix, "BINNED_BLOCK_COLS", 256)
    monkeypatch.setattr(COOMatrix, "BINNED_TILE_ROWS", 512)
    rng = np.random.default_rng(41)
    n, per = 6000, 12
    row = np.repeat(np.arange(n), per).astype(np.int64)
    col = rng.integers(0, n, size=n * per).astype(np.int64)
    col[::per] = np.arange(n)
    val = rng.uniform(-1, 1, size=n * per)
    val[::per] = 2.0 * per
    rhs = rng.standard_normal(n)
    so = S.BiCGStabSolver(hc.oracle_planner("coo", None, row, col, val, n, 1, np.float64, rhs))
    mpl = hc.mirror_planner(gpu_ctx, "coo", None, row, col, val, n, 1, np.float64, rhs)
    matrix = mpl.row_partitioned_matrices[0].matrix
    assert matrix.binned is not None
    sm = BiCGStabSolver(mpl)
    for step in range(6):
        if step == 3:
            assert matrix.release_optional_memory() and matrix.binned is None
            assert not matrix.release_optional_memory()      # nothing else it can work without
        so.step()
        sm.step()
    np.testing.assert_allclose([s.get_value() for s in sm.rho], [float(v) for v in so.rho], rtol=1e-9)
    want = so.planner.get_vector(0, 0)
    got = mpl.get_vector(0, 0).to_host().numpy()
    assert np.abs(got - want).max() <= 1e-9 * np.abs(want).max()


@pytest.mark.parametrize("kind,env", [("csr", {}), ("csr", {"LSK_SLICED": "0"}),
                                      ("csr", {"LSK_SLICED": "0", "LSK_COMPRESS": "0"}), ("coo", {}),
                                      ("coo", {"LSK_SLICED": "0"})])
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_non_finite_elements_of_x_reach_only_the_rows_that_read_them(gpu_ctx, monkeypatch, kind, env, dtype):
    """A NaN and an infinity in x: the rows whose entries multiply them become non-finite exactly where the CPU
    loop's do, every other row keeps its value -- the padding of the set-up-time copies (zeros in the sliced
    copy, clamped gathers in the general kernels) must never multiply an element its row does not read
    (src/CSRMatrixTasks.cpp:73-91, src/COOMatrixTasks.cpp:67-74 touch x only through the entries)."""
    for key, value in env.items():
        monkeypatch.setenv(key, value)
    rects, row, col, val, n = hc.stencil_problem(4, (14, 9, 11), dtype)
    x = np.random.default_rng(37).standard_normal(n).astype(dtype)
    x[[5, n // 2]] = np.nan
    x[[n // 3, n - 2]] = np.inf
    x[0] = -np.inf
    opl = hc.oracle_planner(kind, rects, row, col, val, n, 3, dtype, x)
    mpl = hc.mirror_planner(gpu_ctx, kind, rects, row, col, val, n, 3, dtype, x)
    opl.allocate_workspace(1)
    mpl.allocate_workspace(1)
    opl.matvec(2, 1)
    mpl.matvec(2, 1)
    want = np.asarray(opl.get_vector(2, 0))
    got = mpl.get_vector(2, 0).to_host().numpy()
    assert 20 < int((~np.isfinite(want)).sum()) < n // 2
    assert np.array_equal(np.isnan(got), np.isnan(want))
    assert np.array_equal(np.isposinf(got), np.isposinf(want)) and np.array_equal(np.isneginf(got), np.isneginf(want))
    fine = np.isfinite(want)
    tol = 1e-12 if dtype == np.float64 else 1e-4
    assert np.abs(got[fine] - want[fine]).max() <= tol * np.abs(want[fine]).max()


def test_non_finite_elements_of_x_through_the_binned_copy_and_long_rows(gpu_ctx, monkeypatch):
    """The same for the binned COO copy (whose bins are padded to whole units) and for a CSR matrix with rows
    on the long-row plan (whose chunks are padded to whole waves)."""
    from legionsolvers_amd import COOMatrix
    monkeypatch.setenv("LSK_COO_BLOCK", "1")
    monkeypatch.setenv("LSK_COO_BINNED", "2")
    monkeypatch.setattr(COOMatrix, "BLOCK_MIN_WINDOW", 0)
    monkeypatch.setattr(COOMatrix, "BINNED_BLOCK_COLS", 256)
    monkeypatch.setattr(COOMatrix, "BINNED_TILE_ROWS", 512)
    rng = np.random.default_rng(43)
    n, per = 6000, 12
    row = np.repeat(np.arange(n), per).astype(np.int64)
    col = rng.integers(0, n, size=n * per).astype(np.int64)
    col[::per] = np.arange(n)
    val = rng.uniform(-1, 1, size=n * per)
    x = rng.standard_normal(n)
    x[[7, n // 2]] = np.nan
    x[n - 3] = np.inf
    opl = hc.oracle_planner("coo", None, row, col, val, n, 1, np.float64, x)
    mpl = hc.mirror_planner(gpu_ctx, "coo", None, row, col, val, n, 1, np.float64, x)
    assert mpl.row_partitioned_matrices[0].matrix.binned is not None
    opl.allocate_workspace(1)
    mpl.allocate_workspace(1)
    opl.matvec(2, 1)
    mpl.matvec(2, 1)
    want = np.asarray(opl.get_vector(2, 0))
    got = mpl.get_vector(2, 0).to_host().numpy()
    assert 10 < int((~np.isfinite(want)).sum()) < n // 2
    assert np.array_equal(np.isnan(got), np.isnan(want))
    assert np.array_equal(np.isinf(got), np.isinf(want))
    fine = np.isfinite(want)
    assert np.abs(got[fine] - want[fine]).max() <= 1e-12 * np.abs(want[fine]).max()
    # CSR: 300 short rows and two rows of 20 000 and 9 000 entries (beyond the plan's 8192)
    lens = np.full(300, 5, dtype=np.int64)
    lens[[40, 211]] = (20000, 9000)
    n = 24000
    offs = np.concatenate(([0], np.cumsum(lens)))
    rects = np.stack([offs[:-1], offs[1:] - 1], axis=1).astype(np.int64)
    col = np.concatenate([np.sort(rng.choice(n, size=int(k), replace=False)) for k in lens]).astype(np.int64)
    val = rng.uniform(-1, 1, size=col.size)
    x = rng.standard_normal(n)
    x[[11, n // 2, n - 1]] = (np.nan, np.inf, -np.inf)
    rows = lens.size
    be = gpu_ctx.backend
    d = lambda a: hc.to_dev(gpu_ctx, a)  # noqa: E731
    rowoff = be.csr_compress_rows(d(rects), 0, col.size)
    col32 = be.csr_compress_cols(d(col), 0, n - 1)
    plan = be.csr_long_rows_create(rowoff)
    assert plan is not None
    y = torch.zeros(rows, dtype=torch.float64, device="cuda")
    be.csr_spmv_split(plan, 0, rows, col.size, rowoff, col32, d(val), d(x), 0, n - 1, y, 0)
    want = np.zeros(rows)
    oracle.csr_matvec(rects, 0, col, val, x, 0, n - 1, want, beta=1)
    got = y.cpu().numpy()
    assert np.array_equal(np.isnan(got), np.isnan(want)) and np.array_equal(np.isinf(got), np.isinf(want))
    fine = np.isfinite(want)
    assert fine.sum() > 250 and np.abs(got[fine] - want[fine]).max() <= 1e-12 * np.abs(want[fine]).max()


@pytest.mark.parametrize("switch", ["1", "0", "too-small"])
def test_workspace_vectors_are_placed_by_timing(gpu_ctx, monkeypatch, switch):
    """A matrix with a large sliced-pattern copy: before the planner allocates its workspace it times one product
    into a vector here, into one behind a block of a third of the device and into one behind two (a read and a
    write stream are faster in different thirds of HBM), and allocates the workspace where that was fastest; the
    blocks are gone afterwards, and CG follows the oracle wherever the vectors went.  LSK_PLACE_VECTORS=0: nothing
    is timed; nor is it for a copy of less than PLACE_MIN_BYTES of values (this one has 0.47 GB: the threshold is
    lowered for the first two cases)."""
    from legionsolvers_amd.square_planner import SquarePlanner
    monkeypatch.setenv("LSK_PLACE_VECTORS", "0" if switch == "0" else "1")
    if switch != "too-small":
        monkeypatch.setattr(SquarePlanner, "PLACE_MIN_BYTES", 1 << 20)
    rects, row, col, val, n = hc.stencil_problem(4, (130, 130, 130), np.float64)
    rhs = np.random.default_rng(38).standard_normal(n)
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    free_before, _ = torch.cuda.mem_get_info()
    mpl = hc.mirror_planner(gpu_ctx, "csr", rects, row, col, val, n, 1, np.float64, rhs)
    sm = CGSolver(mpl)
    placed = mpl.workspace_placement
    if switch != "1":
        assert placed is None
    else:
        assert placed["blocks_in_front"] in (0, 1, 2)
        assert 1 <= len(placed["product_ms_by_blocks_in_front"]) <= 3 and min(placed["product_ms_by_blocks_in_front"]) > 0
        torch.cuda.synchronize()
        torch.cuda.empty_cache()
        assert free_before - torch.cuda.mem_get_info()[0] < (8 << 30)    # the blocks (96 GB each) were given back
    so = S.CGSolver(hc.oracle_planner("csr", rects, row, col, val, n, 1, np.float64, rhs))
    for _ in range(5):
        so.step()
        sm.step()
    np.testing.assert_allclose([s.get_value() for s in sm.residual_norm_squared],
                               [float(v) for v in so.residual_norm_squared], rtol=1e-10)


def test_structure_that_cannot_be_compressed_falls_back(gpu_ctx):
    """Rects with a gap (every row ignores its last entry): no 32-bit copy is made, products take
    the reference arrays, CG still follows the oracle."""
    rects, row, col, val, n = hc.stencil_problem(3, (9, 8, 7), np.float64)
    rects = rects.copy()
    rects[rects[:, 1] > rects[:, 0], 1] -= 1
    rhs = np.random.default_rng(34).standard_normal(n)
    opl = hc.oracle_planner("csr", rects, row, col, val, n, 2, np.float64, rhs)
    mpl = hc.mirror_planner(gpu_ctx, "csr", rects, row, col, val, n, 2, np.float64, rhs)
    matrix = mpl.row_partitioned_matrices[0].matrix
    assert matrix.col32 is None and not matrix.rowoff
    so, sm = S.CGSolver(opl), CGSolver(mpl)
    for _ in range(5):
        so.step()
        sm.step()
    np.testing.assert_allclose([s.get_value() for s in sm.residual_norm_squared],
                               [float(v) for v in so.residual_norm_squared], rtol=1e-10)


def test_held_back_solution_update(gpu_ctx, monkeypatch):
    """CG's x += alpha p is held back and shares one pass with p <- r + beta p; reading the solution
    in between (or at any time) gives the same values as with the fused leaves off, bit for bit.
    (The fused product + dot is kept out of it: it sums p.Ap in another order.)"""
    from legionsolvers_amd import Context
    monkeypatch.setenv("LSK_FUSE_SPMV_DOT", "0")
    rects, row, col, val, n = hc.stencil_problem(4, (12, 10, 14), np.float64)
    rhs = np.random.default_rng(35).standard_normal(n)
    runs = []
    for fuse in (True, False):
        ctx = Context(gpu_ctx.device, fuse=fuse)
        mpl = hc.mirror_planner(ctx, "csr", rects, row, col, val, n, 2, np.float64, rhs)
        solver = CGSolver(mpl)
        seen = []
        for step in range(6):
            solver.step()
            if step % 2 == 1:
                seen.append(mpl.get_vector(0, 0).to_host().clone())
        runs.append((seen, mpl._deferred_issued[0], mpl._deferred_fused[0]))
    (fused_seen, issued, won), (plain_seen, issued0, _) = runs
    assert issued == 6 and won == 6 and issued0 == 0
    for a, b in zip(fused_seen, plain_seen):
        assert torch.equal(a, b)
    # a sequence that never continues with the xpay: the bet is dropped after a few losses
    ctx = Context(gpu_ctx.device, fuse=True)
    mpl = hc.mirror_planner(ctx, "csr", rects, row, col, val, n, 2, np.float64, rhs)
    solver = BiCGStabSolver(mpl)
    for _ in range(6):
        solver.step()
    assert mpl._deferred_fused[0] == 0 and mpl._deferred_issued[0] <= 5


def test_column_blocked_coo_follows_the_oracle(gpu_ctx, monkeypatch):
    """An unstructured row-sorted COO matrix whose entries the planner re-orders by column blocks at
    set-up (forced here; normally probed and timed): BiCGStab still follows the oracle."""
    from legionsolvers_amd import COOMatrix
    monkeypatch.setenv("LSK_COO_BLOCK", "1")
    monkeypatch.setenv("LSK_COO_BINNED", "0")
    monkeypatch.setattr(COOMatrix, "BLOCK_MIN_WINDOW", 0)
    monkeypatch.setattr(COOMatrix, "BLOCK_BYTES", 1 << 12)
    rng = np.random.default_rng(36)
    n, per = 6000, 12
    row = np.repeat(np.arange(n), per).astype(np.int64)
    col = rng.integers(0, n, size=n * per).astype(np.int64)
    col[::per] = np.arange(n)                      # a dominant diagonal keeps the system well posed
    val = rng.uniform(-1, 1, size=n * per)
    val[::per] = 2.0 * per
    rhs = rng.standard_normal(n)
    opl = hc.oracle_planner("coo", None, row, col, val, n, 1, np.float64, rhs)
    mpl = hc.mirror_planner(gpu_ctx, "coo", None, row, col, val, n, 1, np.float64, rhs)
    matrix = mpl.row_partitioned_matrices[0].matrix
    assert matrix.blocked and matrix.val_blocked is not None
    so, sm = S.BiCGStabSolver(opl), BiCGStabSolver(mpl)
    for _ in range(5):
        so.step()
        sm.step()
    np.testing.assert_allclose([s.get_value() for s in sm.rho], [float(v) for v in so.rho], rtol=1e-9)
    want = opl.get_vector(0, 0)
    got = mpl.get_vector(0, 0).to_host().numpy()
    assert np.abs(got - want).max() <= 1e-9 * np.abs(want).max()


@pytest.mark.parametrize("solver_name", ["cg", "bicgstab", "gmres"])
@pytest.mark.parametrize("pieces", [1, 3])
def test_two_systems_on_one_matrix_share_a_pass(gpu_ctx, monkeypatch, solver_name, pieces):
    """The reference's BenchmarkStencil solves two systems on one matrix (test/BenchmarkStencil.cpp:
    181-199).  The planner multiplies both in one pass over the matrix values (lsk_csr_sliced_spmv2_*):
    every scalar of the history and both solutions are bit for bit those of LSK_PAIR=0, where the
    two products are launched one after the other; and the pair path really ran."""
    from legionsolvers_amd import CSRMatrix
    rects, row, col, val, n = hc.stencil_problem(4, (13, 11, 17), np.float64)
    rng = np.random.default_rng(71)
    rhs = rng.standard_normal(n)
    calls = []
    real = CSRMatrix.matvec_pair

    def spy(self, *args, **kwargs):
        served = real(self, *args, **kwargs)
        calls.append(served)
        return served

    monkeypatch.setattr(CSRMatrix, "matvec_pair", spy)
    outcomes = []
    for pair in ("1", "0"):
        monkeypatch.setenv("LSK_PAIR", pair)
        calls.clear()
        pl = hc.mirror_planner(gpu_ctx, "csr", rects, row, col, val, n, pieces, np.float64, rhs, spaces=2)
        pl.get_vector(1, 1).owned.mul_(-0.5)   # the second system gets another right-hand side
        solver = {"cg": CGSolver, "bicgstab": BiCGStabSolver}[solver_name](pl) if solver_name != "gmres" \
            else GMRESSolver(pl, 4)
        for _ in range(5):
            solver.step()
        history = {"cg": lambda: solver.residual_norm_squared, "bicgstab": lambda: solver.rho,
                   "gmres": lambda: [h for col_ in solver.inner_products for h in col_]}[solver_name]()
        outcomes.append(([s.get_value() for s in history], pl.get_vector(0, 0).to_host().clone(),
                         pl.get_vector(0, 1).to_host().clone(), list(calls)))
    (h1, a1, b1, calls1), (h0, a0, b0, calls0) = outcomes
    assert calls1 and all(calls1) and not any(calls0)
    assert h1 == h0 and torch.equal(a1, a0) and torch.equal(b1, b0)
    assert not torch.equal(a1, b1)


@pytest.mark.parametrize("solver_name", ["cg", "bicgstab", "gmres"])
@pytest.mark.parametrize("flag,extent,pieces", [(4, (3, 4, 512), 2), (4, (7, 12, 70), 1), (3, (9, 11, 13), 1)],
                         ids=["27-point-2-pieces", "27-point", "7-point"])
def test_uniform_coefficients_give_the_same_bits(gpu_ctx, solver_name, flag, extent, pieces):
    """Option sliced.uniform = 1 on a constant-coefficient stencil, two systems: the products take the
    coefficients as scalars and read no value array, and every scalar of the history and both solutions
    are bit for bit those of the default."""
    import ctypes
    from legionsolvers_amd import _lib
    rects, row, col, val, n = hc.stencil_problem(flag, extent, np.float64)
    rhs = np.random.default_rng(5).standard_normal(n)
    outcomes = []
    try:
        for uniform in (1, 0):
            _lib.call("lsk_set_option", b"sliced.uniform", uniform)
            pl = hc.mirror_planner(gpu_ctx, "csr", rects, row, col, val, n, pieces, np.float64, rhs, spaces=2)
            pl.get_vector(1, 1).owned.mul_(-0.5)
            matrix = pl.row_partitioned_matrices[0].matrix
            assert matrix.sliced and all(s.uniform_values for s in matrix.sliced.values())
            solver = {"cg": CGSolver, "bicgstab": BiCGStabSolver}[solver_name](pl) if solver_name != "gmres" \
                else GMRESSolver(pl, 4)
            for _ in range(5):
                solver.step()
            history = {"cg": lambda: solver.residual_norm_squared, "bicgstab": lambda: solver.rho,
                       "gmres": lambda: [h for col_ in solver.inner_products for h in col_]}[solver_name]()
            outcomes.append(([s.get_value() for s in history], pl.get_vector(0, 0).to_host().clone(),
                             pl.get_vector(0, 1).to_host().clone()))
    finally:
        _lib.call("lsk_set_option", b"sliced.uniform", 0)
    (h1, a1, b1), (h0, a0, b0) = outcomes
    assert h1 == h0 and torch.equal(a1, a0) and torch.equal(b1, b0)


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("shuffled", [False, True])
def test_binned_coo_follows_the_oracle(gpu_ctx, monkeypatch, dtype, shuffled):
    """An unstructured COO matrix multiplied through the binned copy (forced here; normally timed
    against the other forms at set-up), with blocks and tiles small enough that the matrix has
    dozens of each: BiCGStab follows the oracle, and the planner drops the zero fill in front of the
    product (the second pass writes every row)."""
    from legionsolvers_amd import COOMatrix
    monkeypatch.setenv("LSK_COO_BLOCK", "1")
    monkeypatch.setenv("LSK_COO_BINNED", "2")
    monkeypatch.setattr(COOMatrix, "BLOCK_MIN_WINDOW", 0)
    monkeypatch.setattr(COOMatrix, "BINNED_BLOCK_COLS", 256)
    monkeypatch.setattr(COOMatrix, "BINNED_TILE_ROWS", 512)
    rng = np.random.default_rng(37)
    n, per = 6000, 12
    row = np.repeat(np.arange(n), per).astype(np.int64)
    col = rng.integers(0, n, size=n * per).astype(np.int64)
    col[::per] = np.arange(n)                      # a dominant diagonal keeps the system well posed
    val = rng.uniform(-1, 1, size=n * per).astype(dtype)
    val[::per] = 2.0 * per
    if shuffled:
        order = rng.permutation(val.size)
        row, col, val = row[order], col[order], val[order]
    rhs = rng.standard_normal(n).astype(dtype)
    print(f"case binned coo {np.dtype(dtype).name} shuffled={shuffled}", flush=True)
    opl = hc.oracle_planner("coo", None, row, col, val, n, 1, dtype, rhs)
    mpl = hc.mirror_planner(gpu_ctx, "coo", None, row, col, val, n, 1, dtype, rhs)
    matrix = mpl.row_partitioned_matrices[0].matrix
    assert matrix.binned is not None and not matrix.blocked
    assert matrix.binned.blocks == -(-n // 256) and matrix.binned.tiles == -(-n // 512)
    so, sm = S.BiCGStabSolver(opl), BiCGStabSolver(mpl)
    for _ in range(5):
        so.step()
        sm.step()
    rtol = 1e-9 if dtype == np.float64 else 2e-3
    # (in fp32 the later rho are differences of nearly equal numbers six orders below the largest:
    # an absolute bar of a few ulps of that one beside the relative bar)
    rho_want = [float(v) for v in so.rho]
    np.testing.assert_allclose([s.get_value() for s in sm.rho], rho_want, rtol=rtol,
                               atol=0.0 if dtype == np.float64 else 2e-6 * max(abs(v) for v in rho_want))
    want = opl.get_vector(0, 0)
    got = mpl.get_vector(0, 0).to_host().numpy()
    assert np.abs(got - want).max() <= rtol * np.abs(want).max()


@pytest.mark.parametrize("kind", ["csr", "coo"])
@pytest.mark.parametrize("pieces", [1, 3])
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_transposed_product_matches_oracle(gpu_ctx, kind, pieces, dtype, monkeypatch):
    """SquarePlanner.rmatvec (CSRRmatvecTask / COORmatvecTask, assert(false) in the reference:
    parity unpinned there) on an unsymmetric 27-point matrix against the oracle's entry-by-entry
    loop; a second matrix into the same space accumulates; <A A^T y, y> = <A^T y, A^T y>."""
    print(f"case rmatvec {kind} pieces={pieces} {np.dtype(dtype).name}", flush=True)
    rects, row, col, val, n = hc.stencil_problem(4, (9, 7, 8), dtype=dtype)
    rng = np.random.default_rng(11)
    val = rng.standard_normal(val.size).astype(dtype)
    y = rng.standard_normal(n).astype(dtype)
    pl = hc.mirror_planner(gpu_ctx, kind, rects, row, col, val, n, pieces, dtype, y)
    pl.allocate_workspace(2)
    pl.rmatvec(2, 1)
    got = pl.get_vector(2, 0).to_host().numpy()
    want = np.zeros(n, dtype)
    oracle.csr_rmatvec(rects, 0, col, val, y, 0, want, 0, n - 1)
    tol = 1e-12 if dtype == np.float64 else 2e-5
    np.testing.assert_allclose(got, want, rtol=tol, atol=tol)
    pl.matvec(3, 2)
    lhs, rhs = pl.dot(3, 1).get_value(), pl.dot(2, 2).get_value()
    assert abs(lhs - rhs) <= 50 * tol * abs(rhs)
    again = pl.get_vector(2, 0).to_host().numpy()
    pl.rmatvec(2, 1)  # deterministic: the same bits on every call
    assert np.array_equal(pl.get_vector(2, 0).to_host().numpy(), again)
    # transposed in several runs of entries: the runs' products accumulate to the same sums
    from legionsolvers_amd.matrices import TransposedBlock
    monkeypatch.setattr(TransposedBlock, "CHUNK", 157)
    pl.row_partitioned_matrices[0].matrix._transposed = None
    pl.rmatvec(2, 1)
    assert len(pl.row_partitioned_matrices[0].matrix._transposed.parts) > 3
    np.testing.assert_allclose(pl.get_vector(2, 0).to_host().numpy(), want, rtol=tol, atol=tol)
    pl.add_row_partitioned_matrix(pl.row_partitioned_matrices[0].matrix, 0, 0)
    pl.rmatvec(2, 1)
    np.testing.assert_allclose(pl.get_vector(2, 0).to_host().numpy(), 2 * want, rtol=tol, atol=2 * tol)


def test_transposed_product_of_a_rectangular_window(gpu_ctx):
    """The leaf level of rmatvec on a random matrix whose columns partly fall outside the output
    window and whose kernel holds entries that no rect covers: both are dropped, as the oracle
    drops them; the helper arrays are bit-exact."""
    from legionsolvers_amd.matrices import CSRMatrix
    be = gpu_ctx.backend
    rng = np.random.default_rng(3)
    rows, ncols, k_base, row_lo = 3000, 5000, 17, 40
    counts = rng.integers(0, 9, rows)
    gaps = rng.integers(0, 2, rows)          # stray entries between rects
    lo = k_base + np.cumsum(np.concatenate([[0], (counts + gaps)[:-1]]))
    rects = np.stack([lo, lo + counts - 1], axis=1).astype(np.int64)
    nnz = int((counts + gaps).sum())
    col = rng.integers(0, ncols, nnz).astype(np.int64)
    val = rng.standard_normal(nnz)
    x = rng.standard_normal(rows)
    w_lo, w_hi = 700, 4100
    print(f"case rmatvec leaf rows={rows} nnz={nnz} window={w_lo}..{w_hi}", flush=True)
    row32 = be.csr_entry_rows(hc.to_dev(gpu_ctx, rects), k_base, nnz).cpu().numpy().view(np.uint32)
    ref_rows = oracle.csr_entry_rows(rects, k_base, nnz)
    assert np.array_equal(row32, np.where(ref_rows < 0, 0xFFFFFFFF, ref_rows).astype(np.uint32))
    keys = np.sort(rng.integers(0, 50, 1000)).astype(np.uint32)
    keys[-7:] = 0xFFFFFFFF
    off = be.sorted_key_offsets(hc.to_dev(gpu_ctx, keys.view(np.int32)), 60).cpu().numpy().view(np.uint32)
    assert np.array_equal(off, np.searchsorted(keys.astype(np.int64), np.arange(61), side="left"))
    m = CSRMatrix(gpu_ctx, hc.to_dev(gpu_ctx, rects), k_base, hc.to_dev(gpu_ctx, col), hc.to_dev(gpu_ctx, val),
                  IndexSpace(0, row_lo + rows - 1), row_lo)
    out = torch.full((w_hi - w_lo + 1,), 0.5, dtype=torch.float64, device=gpu_ctx.device)
    want = np.full(w_hi - w_lo + 1, 0.5)
    for beta in (1, 0):
        m.rmatvec(out, w_lo, w_hi, hc.to_dev(gpu_ctx, x), row_lo, beta)
        oracle.csr_rmatvec(rects, k_base, col, val, x, row_lo, want, w_lo, w_hi, beta=beta)
        np.testing.assert_allclose(out.cpu().numpy(), want, rtol=1e-12, atol=1e-13)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("pieces", [1, 3])
def test_sliced_pattern_structure_is_used_and_bit_exact(gpu_ctx, monkeypatch, dtype, pieces):
    """A stencil matrix gets the sliced-pattern structure at registration; the planner's product on
    it equals the oracle's ascending-k loop bit for bit, and CG follows the oracle's history."""
    rects, row, col, val, n = hc.stencil_problem(4, (11, 9, 21), dtype)
    val = np.random.default_rng(40).standard_normal(val.size).astype(dtype)
    rhs = np.random.default_rng(41).standard_normal(n).astype(dtype)
    print(f"case sliced planner {np.dtype(dtype).name} pieces={pieces}", flush=True)
    mpl = hc.mirror_planner(gpu_ctx, "csr", rects, row, col, val, n, pieces, dtype, rhs)
    matrix = mpl.row_partitioned_matrices[0].matrix
    assert sorted(matrix.sliced) == list(range(pieces)) and matrix.fuses_dot()
    mpl.allocate_workspace(2)
    mpl.matvec(2, 1)
    want = np.zeros(n, dtype)
    oracle.csr_matvec(rects, 0, col, val, rhs, 0, n - 1, want)
    assert np.array_equal(mpl.get_vector(2, 0).to_host().numpy(), want)
    # the fused product + dot between workspace vectors: the same product, and the dot the planner
    # then takes from its shares
    mpl.copy(2, 1)
    mpl.matvec(3, 2)
    assert np.array_equal(mpl.get_vector(3, 0).to_host().numpy(), want)
    assert mpl._fused_dot_pending is not None
    got = mpl.dot(2, 3).get_value()
    exact = float(np.dot(want.astype(np.float64), rhs.astype(np.float64)))
    scale = float(np.dot(np.abs(want).astype(np.float64), np.abs(rhs).astype(np.float64)))
    assert mpl._fused_dot_used == 1 and abs(got - exact) <= (1e-13 if dtype == np.float64 else 1e-5) * scale
    monkeypatch.setenv("LSK_SLICED", "0")
    mpl0 = hc.mirror_planner(gpu_ctx, "csr", rects, row, col, val, n, pieces, dtype, rhs)
    assert not mpl0.row_partitioned_matrices[0].matrix.sliced
    mpl0.allocate_workspace(1)
    mpl0.matvec(2, 1)
    tol = 1e-12 if dtype == np.float64 else 1e-4
    np.testing.assert_allclose(mpl0.get_vector(2, 0).to_host().numpy(), want, rtol=tol, atol=tol)


@pytest.mark.parametrize("pieces", [1, 3])
def test_row_sorted_coo_stencil_gets_the_sliced_copy(gpu_ctx, monkeypatch, pieces):
    """A row-sorted COO matrix of a stencil is multiplied through the same sliced-pattern copy as
    its CSR form: bit for bit the oracle's entry-by-entry COO loop; a shuffled one keeps the scan
    kernel; CG follows the oracle either way."""
    from legionsolvers_amd import COOMatrix
    rects, row, col, val, n = hc.stencil_problem(4, (9, 10, 13), np.float64)
    val = np.random.default_rng(50).standard_normal(val.size)
    rhs = np.random.default_rng(51).standard_normal(n)
    print(f"case coo sliced pieces={pieces}", flush=True)
    mpl = hc.mirror_planner(gpu_ctx, "coo", rects, row, col, val, n, pieces, np.float64, rhs)
    matrix = mpl.row_partitioned_matrices[0].matrix
    assert isinstance(matrix, COOMatrix) and sorted(matrix.sliced) == list(range(pieces))
    mpl.allocate_workspace(1)
    mpl.matvec(2, 1)
    want = np.zeros(n)
    oracle.coo_matvec(row, col, val, rhs, 0, n - 1, want, 0, n - 1)
    assert np.array_equal(mpl.get_vector(2, 0).to_host().numpy(), want)
    # the symmetric positive definite stencil itself, entries shuffled: no sliced copy, same CG
    rects, row, col, val, n = hc.stencil_problem(4, (9, 10, 13), np.float64)
    perm = np.random.default_rng(52).permutation(val.size)
    opl = hc.oracle_planner("coo", rects, row, col, val, n, pieces, np.float64, rhs)
    histories = []
    for r, c, v in ((row, col, val), (row[perm], col[perm], val[perm])):
        pl = hc.mirror_planner(gpu_ctx, "coo", rects, r, c, v, n, pieces, np.float64, rhs)
        assert bool(pl.row_partitioned_matrices[0].matrix.sliced) == (r is row)
        solver = CGSolver(pl)
        for _ in range(6):
            solver.step()
        histories.append([s.get_value() for s in solver.residual_norm_squared])
    so = S.CGSolver(opl)
    for _ in range(6):
        so.step()
    for h in histories:
        np.testing.assert_allclose(h, [float(v) for v in so.residual_norm_squared], rtol=1e-10)


@pytest.mark.parametrize("kind,sliced", [("csr", "1"), ("csr", "0"), ("coo", "1"), ("coo", "blocked"), ("coo", "binned")])
def test_values_changed_brings_every_copy_up_to_date(gpu_ctx, monkeypatch, kind, sliced):
    """Set-up may move the arrays (placement redraws) and keeps copies of the values (sliced-pattern
    copy, column-blocked COO order, the transpose); the reference's tasks read the region on every call
    (src/CSRMatrixTasks.cu:88-118).  After the caller rewrites `val` in place and says so, every product
    equals the oracle's on the new values."""
    from legionsolvers_amd import COOMatrix, CSRMatrix
    monkeypatch.setattr(CSRMatrix, "TUNE_MIN_ROWS", 1)   # force the tuner and the redraws on a small problem
    monkeypatch.setattr(COOMatrix, "TUNE_MIN_ROWS", 1)
    monkeypatch.setenv("LSK_TUNE", "2")   # (the redraws are opt-in)
    monkeypatch.setenv("LSK_TUNE_REDRAWS", "2")
    rng = np.random.default_rng(60)
    if sliced in ("blocked", "binned"):
        monkeypatch.setenv("LSK_COO_BLOCK", "1")
        monkeypatch.setenv("LSK_COO_BINNED", "2" if sliced == "binned" else "0")
        monkeypatch.setattr(COOMatrix, "BLOCK_MIN_WINDOW", 0)
        monkeypatch.setattr(COOMatrix, "BLOCK_BYTES", 1 << 12)
        monkeypatch.setattr(COOMatrix, "BINNED_BLOCK_COLS", 512)
        monkeypatch.setattr(COOMatrix, "BINNED_TILE_ROWS", 256)
        n, per = 6000, 12
        rects = None
        row = np.repeat(np.arange(n), per).astype(np.int64)
        col = rng.integers(0, n, size=n * per).astype(np.int64)
        pieces = 1
    else:
        monkeypatch.setenv("LSK_SLICED", sliced)
        rects, row, col, _, n = hc.stencil_problem(4, (11, 9, 21), np.float64)
        pieces = 2
    old = rng.standard_normal(col.size)
    new = rng.standard_normal(col.size)
    x = rng.standard_normal(n)
    print(f"case values_changed {kind} sliced={sliced}", flush=True)
    mpl = hc.mirror_planner(gpu_ctx, kind, rects, row, col, old, n, pieces, np.float64, x)
    matrix = mpl.row_partitioned_matrices[0].matrix
    if sliced == "blocked":
        assert matrix.blocked
    elif sliced == "binned":
        assert matrix.binned is not None
    else:
        assert bool(matrix.sliced) == (sliced == "1")
    given = matrix._val_given if kind == "csr" else matrix.val
    mpl.allocate_workspace(2)

    def products():
        mpl.matvec(2, 1)
        mpl.rmatvec(3, 1)
        return mpl.get_vector(2, 0).to_host().numpy(), mpl.get_vector(3, 0).to_host().numpy()

    def expected(values):
        y, yt = np.zeros(n), np.zeros(n)
        oracle.coo_matvec(row, col, values, x, 0, n - 1, y, 0, n - 1)
        oracle.coo_matvec(col, row, values, x, 0, n - 1, yt, 0, n - 1)   # the transpose, entry by entry
        return y, yt

    for got, want in zip(products(), expected(old)):
        np.testing.assert_allclose(got, want, rtol=1e-12, atol=1e-12)
    given.copy_(hc.to_dev(gpu_ctx, new))   # the caller rewrites its entries in place ...
    matrix.values_changed()                # ... and says so
    for got, want in zip(products(), expected(new)):
        np.testing.assert_allclose(got, want, rtol=1e-12, atol=1e-12)


def test_write_through_a_view_is_seen_by_the_next_dot(gpu_ctx):
    """A fused leaf leaves partial sums of a dot product with the vector; a write through one of the
    vector's views afterwards must not be answered with those stale sums."""
    n = 70000
    ip = IndexPartition.equal(IndexSpace(0, n - 1), 3)
    g = torch.Generator(device="cuda").manual_seed(9)
    x = PartitionedVector(gpu_ctx, "x", ip)
    y = PartitionedVector(gpu_ctx, "y", ip)
    x.owned.copy_(torch.randn(n, generator=g, device="cuda", dtype=torch.float64))
    y.owned.copy_(torch.randn(n, generator=g, device="cuda", dtype=torch.float64))
    y.axpy(0.5, x, speculate=True)          # leaves the shares of y.y
    assert y._ready is not None
    first = y.dot(y).get_value()
    y.owned.mul_(2.0)                        # an external write through the view
    assert y._ready is None
    assert y.dot(y).get_value() == pytest.approx(4.0 * first, rel=1e-13)
    kept = y.piece(1)
    y.axpy(0.25, x, speculate=True)
    kept.zero_()                             # a view kept across an operation: the caller says so
    y.mark_written()
    fresh = PartitionedVector(gpu_ctx, "f", ip)
    fresh.assign(y)
    assert y.dot(y).get_value() == fresh.dot(fresh).get_value()


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("pieces", [1, 3])
def test_classical_gram_schmidt_gmres_follows_its_oracle(gpu_ctx, dtype, pieces):
    """GMRES with the classical Gram-Schmidt option (one pass over w for all inner products of a
    sweep, one for all updates) against the oracle's restatement of that variant; parity unpinned by
    the reference, which has only the modified sweep."""
    rects, row, col, val, n = hc.stencil_problem(3, (12, 10, 11), dtype)
    rhs = np.random.default_rng(70).standard_normal(n).astype(dtype)
    print(f"case classical GMRES {np.dtype(dtype).name} pieces={pieces}", flush=True)
    opl = hc.oracle_planner("csr", rects, row, col, val, n, pieces, dtype, rhs)
    mpl = hc.mirror_planner(gpu_ctx, "csr", rects, row, col, val, n, pieces, dtype, rhs)
    so, sm = S.GMRESSolver(opl, 12, "classical"), GMRESSolver(mpl, 12, "classical")
    for _ in range(2):
        so.step()
        sm.step()
    tol = 1e-9 if dtype == np.float64 else 2e-3
    want_h = np.array([[float(v) for v in r] for r in so.inner_products])
    got_h = np.array([[s.get_value() for s in r] for r in sm.inner_products])
    assert np.abs(got_h - want_h).max() <= tol * np.abs(want_h).max()
    want = opl.get_vector(0, 0).astype(np.float64)
    got = mpl.get_vector(0, 0).to_host().numpy().astype(np.float64)
    assert np.abs(got - want).max() <= tol * np.abs(want).max()


def test_fused_dot_bets_come_back_after_plain_products(gpu_ctx):
    """The planner fuses the shares of a dot product into a product as a bet on the caller's next dot; a run
    of products that nobody follows with a dot stops the bets, and a solver that comes afterwards gets them
    back within a few steps (CG's p . A p wins every one)."""
    rects, row, col, val, n = hc.stencil_problem(4, (7, 12, 70), np.float64)
    pl = hc.mirror_planner(gpu_ctx, "csr", rects, row, col, val, n, 1, np.float64, np.ones(n))
    solver = CGSolver(pl)
    for _ in range(24):
        pl.matvec(CGSolver.Q, CGSolver.P)
    assert pl._fused_dot_used == 0 and pl._fused_dot_issued >= 3   # bets were lost, and are placed sparingly now
    used = pl._fused_dot_used
    for _ in range(40):
        solver.step()
    assert pl._fused_dot_used - used >= 15, (used, pl._fused_dot_used, pl._fused_dot_issued)


@pytest.mark.parametrize("pieces", [1, 3])
def test_planner_splits_rows_too_long_for_one_wave(gpu_ctx, monkeypatch, pieces):
    """A matrix with a power-law row-length distribution through the planner: the pieces that hold rows of
    tens of thousands of entries get the long-row plan at set-up, every product goes through the split
    product, and CG-shaped planner calls (product, dot, axpy) follow the oracle; LSK_SPLIT_ROWS=0 gives the
    plain product and the same values to rounding."""
    rng = np.random.default_rng(71)
    n = 2400
    lens = np.minimum((10.0 * rng.random(n) ** (-1.0 / 1.5)).astype(np.int64), n)
    lens[[5, 1300, 2399]] = [2400, 2100, 1999]
    lens = np.repeat(lens, 1)
    # rows of up to 2400 entries are not "long" for the library's 8192: tile the pattern to make them so
    reps = 5
    starts = np.concatenate([[0], np.cumsum(lens * reps)])
    rects = np.stack([starts[:-1], starts[1:] - 1], axis=1).astype(np.int64)
    nnz = int(starts[-1])
    col = rng.integers(0, n, size=nnz).astype(np.int64)
    val = rng.standard_normal(nnz)
    row = np.repeat(np.arange(n, dtype=np.int64), lens * reps)
    rhs = rng.standard_normal(n)
    print(f"case planner long rows pieces={pieces} nnz={nnz} longest={int((lens * reps).max())}", flush=True)
    results = []
    for split in ("1", "0"):
        monkeypatch.setenv("LSK_SPLIT_ROWS", split)
        opl = hc.oracle_planner("csr", rects, row, col, val, n, pieces, np.float64, rhs)
        mpl = hc.mirror_planner(gpu_ctx, "csr", rects, row, col, val, n, pieces, np.float64, rhs)
        matrix = mpl.row_partitioned_matrices[0].matrix
        assert not matrix.sliced and matrix.col32 is not None
        assert bool(matrix.long_rows) == (split == "1")
        if split == "1":
            assert sum(p.long_rows for p in matrix.long_rows.values()) == int((lens * reps > 8192).sum()) >= 3
            assert not matrix.fuses_dot()
        opl.allocate_workspace(2)
        mpl.allocate_workspace(2)
        for pl in (opl, mpl):
            pl.matvec(2, 1)                  # w2 <- A rhs
            pl.matvec(3, 2)                  # w3 <- A w2
            pl.axpy(3, pl.dot(2, 3), 1)      # w3 += (w2 . w3) rhs
        want = opl.get_vector(3, 0)
        got = mpl.get_vector(3, 0).to_host().numpy()
        assert np.abs(got - want).max() <= 1e-11 * np.abs(want).max()
        results.append(got)
    assert np.abs(results[0] - results[1]).max() <= 1e-11 * np.abs(results[0]).max()
