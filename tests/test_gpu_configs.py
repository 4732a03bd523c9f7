"""The reference's benchmark configurations other than the headline one, as parity cases:

* Test05COOSolveCG at N = 1024 (1-D Laplacian, COO, one shard);
* Test06CSRSolveCG on the 2-D 5-point Laplacian at 4096^2 (CSR, one shard), at FULL size against
  the oracle's sequential loops;
* BiCGStab on a random nonsymmetric row-sorted COO matrix with 32 entries per row (4 shards):
  oracle parity at a reduced row count, and at the full 5e8 entries the COO kernel against the
  CSR kernel over the same arrays;
* GMRES(30) on the 3-D 7-point stencil in fp32 (8 shards): oracle parity on a small grid, and
  size-independent properties at 768^3.

None of these has a golden in the reference (its tests only pin N = 100 CG), so the expected
values come from the oracle's restatement: parity unpinned by the reference, pinned by the
restatement.  fp64 residual histories: 1e-10 relative; integer arrays: bit-exact."""
import numpy as np
import pytest
import torch

import oracle
from oracle import solvers as S

import host_cases as hc

from legionsolvers_amd import (BiCGStabSolver, CGSolver, COOMatrix, CSRMatrix, GMRESSolver,
                               IndexPartition, IndexSpace, PartitionedVector, SquarePlanner)
from legionsolvers_amd import stencil_generator as sg

pytestmark = pytest.mark.gpu


def _cg_planner(ctx, matrix, ip, dtype=torch.float64):
    rhs = PartitionedVector(ctx, "rhs", ip, dtype)
    rhs.constant_fill(1.0)
    sol = PartitionedVector(ctx, "sol", ip, dtype)
    sol.zero_fill()
    planner = SquarePlanner(ctx, dtype)
    planner.add_sol_vector(sol)
    planner.add_rhs_vector(rhs)
    planner.add_row_partitioned_matrix(matrix, 0, 0)
    return planner, sol


def test_cg_coo_laplacian_1024(gpu_ctx):
    """Test05COOSolveCG's system at N = 1024, one shard, 20 CG steps."""
    n, steps = 1024, 20
    print(f"case cg coo 1-D n={n}", flush=True)
    ip = IndexPartition.equal(IndexSpace(0, n - 1), 1)
    planner, sol = _cg_planner(gpu_ctx, sg.coo_negative_laplacian_1d(gpu_ctx, n, ip), ip)
    solver = CGSolver(planner)
    rects, row, col, val = oracle.laplacian_1d(n)
    opl = hc.oracle_planner("coo", rects, row, col, val, n, 1, np.float64)
    osolver = S.CGSolver(opl)
    for _ in range(steps):
        solver.step()
        osolver.step()
    np.testing.assert_allclose([s.get_value() for s in solver.residual_norm_squared],
                               [float(v) for v in osolver.residual_norm_squared], rtol=1e-10)
    np.testing.assert_allclose(sol.to_host().numpy(), opl.get_vector(0, 0), rtol=1e-10)


def test_cg_csr_laplacian_2d_4096_full_size(gpu_ctx):
    """Test06CSRSolveCG's configuration: 2-D 5-point Laplacian, 4096^2 rows, CSR, fp64, one
    shard, b = 1, x0 = 0.  The device generator's arrays are compared bit for bit with the
    oracle's walk, then 10 CG steps against the oracle's sequential loops."""
    nx = 4096
    n, steps = nx * nx, 10
    print(f"case cg csr 2-D {nx}^2 full size", flush=True)
    _, offs, cfs = sg.benchmark_stencil(2)
    ip = IndexPartition.equal(IndexSpace(0, n - 1), 1)
    A = sg.create_linearized_csr_stencil_matrix(gpu_ctx, (nx, nx), offs, cfs, ip)
    ooffs, ocfs = oracle.benchmark_stencil_offsets(2)
    rects, _, col, val = oracle.linearized_stencil((nx, nx), ooffs, ocfs)
    assert A.nnz == col.size == 5 * n - 4 * nx
    assert np.array_equal(A.rects.cpu().numpy(), rects)
    assert np.array_equal(A.col.cpu().numpy(), col)
    assert np.array_equal(A.val.cpu().numpy(), val)

    planner, sol = _cg_planner(gpu_ctx, A, ip)
    solver = CGSolver(planner)
    # CGSolver.hpp:32-55 on one piece, straight on the oracle's leaf loops
    x, b = np.zeros(n), np.ones(n)
    p, r, q = b.copy(), b.copy(), np.zeros(n)
    hist = [oracle.dot(r, r)]
    for _ in range(steps):
        solver.step()
        q[:] = 0
        oracle.csr_matvec(rects, 0, col, val, p, 0, n - 1, q, beta=1)
        pq, rr_old = oracle.dot(p, q), hist[-1]
        oracle.axpy(oracle.get_alpha([rr_old, pq], np.float64), p, x)
        oracle.axpy(oracle.get_alpha([np.float64(-1), rr_old, pq], np.float64), q, r)
        hist.append(oracle.dot(r, r))
        oracle.xpay(oracle.get_alpha([hist[-1], rr_old], np.float64), r, p)
    got = [s.get_value() for s in solver.residual_norm_squared]
    np.testing.assert_allclose(got, [float(v) for v in hist], rtol=1e-10)
    xs = sol.to_host().numpy()
    assert np.abs(xs - x).max() <= 1e-10 * np.abs(x).max()
    # one product as the planner launches it (the sliced-pattern copy) against the CPU loop, bit by bit, every row
    matrix = planner.row_partitioned_matrices[0].matrix
    assert matrix.sliced
    xv = np.random.default_rng(62).standard_normal(n)
    src, dst = planner.get_vector(CGSolver.P, 0), planner.get_vector(CGSolver.Q, 0)
    src.owned.copy_(hc.to_dev(gpu_ctx, xv))
    planner.matvec(CGSolver.Q, CGSolver.P)
    want = np.zeros(n)
    oracle.csr_matvec(rects, 0, col, val, xv, 0, n - 1, want, beta=1)
    differing = int((want.view(np.uint64) != dst.to_host().numpy().view(np.uint64)).sum())
    print(f"full size: the sliced 2-D 5-point product against the CPU loop at {nx}^2: {differing} of {n} rows differ in any bit")
    assert differing == 0


def _random_rowsorted_coo(rows, per_row, seed=20250310):
    """Row-sorted COO with `per_row` entries per row: entry 0 of a row is the diagonal (64),
    the others have uniform columns over [0, rows) and uniform(-1, 1) values (PCG64)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    row = np.repeat(np.arange(rows, dtype=np.int64), per_row)
    col = rng.integers(0, rows, size=rows * per_row, dtype=np.int64)
    val = rng.uniform(-1.0, 1.0, size=rows * per_row)
    col[::per_row] = np.arange(rows, dtype=np.int64)
    val[::per_row] = 64.0
    return row, col, val


def test_bicgstab_random_coo_scaled(gpu_ctx):
    """BiCGStab on the random nonsymmetric COO configuration at 1/125 of its row count
    (125 000 rows, 4.0e6 entries), 4 shards, against the oracle."""
    rows, per_row, pieces, steps = 125_000, 32, 4, 5
    print(f"case bicgstab random coo rows={rows}", flush=True)
    row, col, val = _random_rowsorted_coo(rows, per_row)
    rhs = np.random.Generator(np.random.PCG64(7)).standard_normal(rows)
    opl = hc.oracle_planner("coo", None, row, col, val, rows, pieces, np.float64, rhs)
    mpl = hc.mirror_planner(gpu_ctx, "coo", None, row, col, val, rows, pieces, np.float64, rhs)
    so, sm = S.BiCGStabSolver(opl), BiCGStabSolver(mpl)
    # The system is strongly diagonally dominant: the residual falls by ~1e-2 per step, so
    # after two steps the recurrence scalars are differences of nearly equal numbers and no
    # longer comparable to 1e-10; the histories are compared up to there, the solution after 5.
    for _ in range(2):
        so.step()
        sm.step()
    for name in ("rho", "alpha", "omega"):
        np.testing.assert_allclose([s.get_value() for s in getattr(sm, name)],
                                   [float(v) for v in getattr(so, name)], rtol=1e-10, err_msg=name)
    for _ in range(steps - 2):
        so.step()
        sm.step()
    want = opl.get_vector(0, 0)
    got = mpl.get_vector(0, 0).to_host().numpy()
    assert np.abs(got - want).max() <= 1e-10 * np.abs(want).max()
    y = np.zeros(rows)
    oracle.coo_matvec(row, col, val, got, 0, rows - 1, y, 0, rows - 1)
    assert np.linalg.norm(rhs - y) <= 1e-10 * np.linalg.norm(rhs)


@pytest.mark.parametrize("rows", [15_625_000], ids=["15625000rows"])
def test_coo_against_csr_at_full_random_size(gpu_ctx, rows):
    """The random COO configuration at full size (15 625 000 rows x 32 = 5.0e8 entries, 12 GB):
    every row has exactly 32 consecutive entries, so the same arrays are also a CSR matrix with
    rects {32 r, 32 r + 31}; the COO kernel (atomics) and the CSR kernel must agree."""
    from conftest import require_device_memory
    per_row = 32
    require_device_memory(60, f"random COO (scan kernel and binned copy) against CSR at {rows} rows x {per_row} entries")
    nnz = rows * per_row
    print(f"case coo vs csr random rows={rows} nnz={nnz}", flush=True)
    g = torch.Generator(device="cuda").manual_seed(20250310)
    dev = gpu_ctx.device
    row = torch.arange(rows, device=dev, dtype=torch.int64).repeat_interleave(per_row)
    col = torch.randint(0, rows, (nnz,), generator=g, device=dev, dtype=torch.int64)
    val = torch.rand(nnz, generator=g, device=dev, dtype=torch.float64) * 2 - 1
    col[::per_row] = torch.arange(rows, device=dev, dtype=torch.int64)
    val[::per_row] = 64.0
    lo = torch.arange(rows, device=dev, dtype=torch.int64) * per_row
    rects = torch.stack([lo, lo + (per_row - 1)], dim=1).contiguous()
    space = IndexSpace(0, rows - 1)
    ip = IndexPartition.equal(space, 1)
    coo = COOMatrix(gpu_ctx, row, col, val, space)
    csr = CSRMatrix(gpu_ctx, rects, 0, col, val, space, 0)
    x = PartitionedVector(gpu_ctx, "x", ip)
    x.owned.copy_(torch.randn(rows, generator=g, device=dev, dtype=torch.float64))
    y_coo = PartitionedVector(gpu_ctx, "y_coo", ip)
    y_csr = PartitionedVector(gpu_ctx, "y_csr", ip)
    y_coo.zero_fill()
    kp_coo = coo.create_kernel_partition_from_range_partition(ip)
    kp_csr = csr.create_kernel_partition_from_range_partition(ip)
    assert kp_coo.intervals == kp_csr.intervals == [(0, nnz - 1)]
    coo.matvec(y_coo, x.owned, 0, rows - 1, kp_coo, 1)
    csr.matvec(y_csr, x.owned, 0, rows - 1, kp_csr, 0)
    scale = float(y_csr.owned.abs().max())
    assert scale > 1.0
    assert float((y_coo.owned - y_csr.owned).abs().max()) <= 1e-13 * scale
    # a slice of rows against the oracle's sequential COO loop
    take = 4096 * per_row
    want = np.zeros(4096)
    oracle.coo_matvec(row[:take].cpu().numpy(), col[:take].cpu().numpy(), val[:take].cpu().numpy(),
                      x.owned.cpu().numpy(), 0, rows - 1, want, 0, 4095)
    np.testing.assert_allclose(y_coo.owned[:4096].cpu().numpy(), want, rtol=0, atol=1e-13 * scale)
    # The form the planner picks for this matrix and the benchmark's BiCGStab steps run on: the binned copy
    # (954 blocks of 16384 columns x 768 tiles of 20352 rows, 32-bit positions up to 5e8), beta = 0 and 1,
    # to the same bar; and again after the values were rewritten in place.
    del coo, y_coo
    be = gpu_ctx.backend
    row32, col32 = be.csr_compress_cols(row, 0, rows - 1), be.csr_compress_cols(col, 0, rows - 1)
    del row
    print(f"case binned coo copy at full size rows={rows} nnz={nnz}", flush=True)
    binned = be.coo_binned_create(row32, col32, val, 0, rows - 1, rows)
    assert binned is not None
    try:
        assert binned.live == nnz and binned.blocks == -(-rows // binned.block_cols) and binned.tiles * binned.tile_rows >= rows
        assert binned.padded_entries < 1.01 * nnz and binned.segments <= binned.blocks * binned.tiles
        y_bin = torch.full((rows,), 5.0, dtype=torch.float64, device=dev)
        be.coo_binned_spmv(binned, x.owned, y_bin, 0)
        assert float((y_bin - y_csr.owned).abs().max()) <= 1e-13 * scale
        be.coo_binned_spmv(binned, x.owned, y_bin, 1)
        assert float((y_bin - 2.0 * y_csr.owned).abs().max()) <= 2e-13 * scale
        val.mul_(-0.375)                                   # exact in binary: the products scale exactly
        be.coo_binned_update_values(binned, val)
        be.coo_binned_spmv(binned, x.owned, y_bin, 0)
        assert float((y_bin + 0.375 * y_csr.owned).abs().max()) <= 1e-13 * scale
    finally:
        binned.close()


def test_gmres30_7pt_fp32_scaled(gpu_ctx):
    """GMRES(30) on the 3-D 7-point stencil in fp32, 8 shards, 24^3 grid, two restart cycles,
    against the oracle's fp32 restatement (including the reference's dummy coefficient)."""
    ext, pieces = (24, 24, 24), 8
    print(f"case gmres30 7pt fp32 {ext}", flush=True)
    rects, row, col, val, n = hc.stencil_problem(3, ext, np.float32)
    rhs = np.random.default_rng(11).standard_normal(n)
    opl = hc.oracle_planner("csr", rects, row, col, val, n, pieces, np.float32, rhs)
    mpl = hc.mirror_planner(gpu_ctx, "csr", rects, row, col, val, n, pieces, np.float32, rhs)
    so, sm = S.GMRESSolver(opl, 30), GMRESSolver(mpl, 30)
    for _ in range(2):
        so.step()
        sm.step()
    # Hessenberg entries of the last cycle: fp32 sums in a different order, 30 Gram-Schmidt
    # sweeps deep
    want_h = np.array([[float(v) for v in r] for r in so.inner_products])
    got_h = np.array([[s.get_value() if hasattr(s, "get_value") else float(s) for s in r]
                      for r in sm.inner_products])
    assert np.abs(got_h - want_h).max() <= 1e-3 * np.abs(want_h).max()
    want = opl.get_vector(0, 0).astype(np.float64)
    got = mpl.get_vector(0, 0).to_host().numpy().astype(np.float64)
    assert np.abs(got - want).max() <= 1e-3 * np.abs(want).max()


@pytest.mark.parametrize("n1", [768], ids=["768cubed"])
def test_spmv_properties_7pt_fp32_768(gpu_ctx, monkeypatch, n1):
    """The GMRES configuration's matrix at full size (7-point, 768^3, fp32 entries, int64
    indices: 453M rows, 3.17G entries, 45 GB): entry count, zero interior row sums,
    symmetry and linearity of the fp32 SpMV."""
    from conftest import require_device_memory
    require_device_memory(120, f"7-point fp32 CSR SpMV properties at {n1}^3 (3 167 354 880 entries)")
    n = n1 ** 3
    print(f"case spmv properties 7pt fp32 {n1}^3", flush=True)
    _, offs, cfs = sg.benchmark_stencil(3)
    ip = IndexPartition.equal(IndexSpace(0, n - 1), 1)
    A = sg.create_linearized_csr_stencil_matrix(gpu_ctx, (n1, n1, n1), offs, cfs, ip, torch.float32)
    assert A.nnz == 7 * n - 6 * n1 * n1
    kp = A.create_kernel_partition_from_range_partition(ip)

    def matvec(x, name):
        y = PartitionedVector(gpu_ctx, name, ip, torch.float32)
        A.matvec(y, x.owned, 0, n - 1, kp, 0)
        return y

    ones = PartitionedVector(gpu_ctx, "ones", ip, torch.float32)
    ones.constant_fill(1.0)
    rowsum = matvec(ones, "rowsum").owned.view(n1, n1, n1)
    assert float(rowsum[1:-1, 1:-1, 1:-1].abs().max()) == 0.0  # 6 - 6*1 exactly, any order
    assert float(rowsum[0].min()) >= 1.0 and float(rowsum[:, :, -1].min()) >= 1.0
    del rowsum
    g = torch.Generator(device="cuda").manual_seed(9)
    x = PartitionedVector(gpu_ctx, "x", ip, torch.float32)
    y = PartitionedVector(gpu_ctx, "yv", ip, torch.float32)
    x.owned.copy_(torch.randn(n, generator=g, device="cuda", dtype=torch.float32))
    y.owned.copy_(torch.randn(n, generator=g, device="cuda", dtype=torch.float32))
    Ax, Ay = matvec(x, "Ax"), matvec(y, "Ay")
    # fp64 accumulation for the check itself, so that only the SpMV's fp32 rounding is measured
    xAy = float(torch.dot(x.owned.double(), Ay.owned.double()))
    yAx = float(torch.dot(y.owned.double(), Ax.owned.double()))
    norm = float(torch.linalg.vector_norm(x.owned.double()) * torch.linalg.vector_norm(Ay.owned.double()))
    assert abs(xAy - yAx) <= 1e-6 * norm
    z = PartitionedVector(gpu_ctx, "z", ip, torch.float32)
    z.assign(x)
    z.axpy(1.0, y)
    Az = matvec(z, "Az")
    err = float((Az.owned.double() - Ax.owned.double() - Ay.owned.double()).abs().max())
    assert err <= 4e-6 * float(Ax.owned.abs().max() + Ay.owned.abs().max())
    # the planner's two set-up-time layouts of this matrix against the product on the int64 arrays
    del Az, Ay, z, y
    monkeypatch.setenv("LSK_TUNE_REDRAWS", "0")
    assert A.compress(0, n - 1, kp, ip)
    assert torch.equal(matvec(x, "Ax32").owned, Ax.owned)
    assert A.slice_structure(0, n - 1, kp, ip)
    s = A.sliced[0]
    assert (s.patterns, s.mixed_blocks) == (27, 0) and A.nnz < s.padded_entries <= 1.01 * A.nnz  # rows padded to their block's longest
    Axs = matvec(x, "Axs").owned
    diff = float((Axs - Ax.owned).abs().max())
    assert diff <= 2e-6 * float(Ax.owned.abs().max())
    # ... and the sliced product against the CPU loop itself (src/CSRMatrixTasks.cpp:73-91: every row summed in
    # entry order), bit by bit, all 453 million rows: the matrix goes to the host for it (45 GB)
    got = Axs.cpu().numpy()
    x_host = x.owned.cpu().numpy()
    del Axs, Ax
    rects_h, col_h, val_h = A.rects.cpu().numpy(), A.col.cpu().numpy(), A.val.cpu().numpy()
    want = np.zeros(n, dtype=np.float32)
    oracle.csr_matvec(rects_h, int(A.k_base), col_h, val_h, x_host, 0, n - 1, want, beta=1, mt=True)
    differing = int((want.view(np.uint32) != got.view(np.uint32)).sum())
    print(f"full size: the sliced 7-point fp32 product against the CPU loop at {n1}^3: {differing} of {n} rows differ in any bit")
    assert differing == 0


@pytest.mark.parametrize("graph", [0, 1])
@pytest.mark.parametrize("solver", [1, 2, 3])
def test_benchmark_stencil_cli(solver, graph, capsys):
    """The BenchmarkStencil program with the reference's flags (test/BenchmarkStencil.cpp:36-60)
    and its closing lines (:213-227), CG / BiCGStab / GMRES on a small 27-point problem; -graph 1: its
    traces recorded once and replayed."""
    from legionsolvers_amd import benchmark_stencil as bs
    print(f"case BenchmarkStencil cli solver={solver} graph={graph}", flush=True)
    bs.main(["-dim", "4", "-nx", "16", "-ny", "12", "-nz", "20", "-vp", "4", "-solver", str(solver),
             "-it", "12", "-pt", "2", "-graph", str(graph)] + (["-systems", "1"] if solver == 2 else []))
    out = capsys.readouterr().out
    assert "Performed 5 warmup traces." in out
    assert "Performed 1 timed traces." in out
    assert "ms per iteration." in out
