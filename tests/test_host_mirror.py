"""CPU tests of the host mirror's LOGIC: partitions, halo plans, the planner and the solver
sequences, run with the oracle leaf backend injected (tests/oracle_backend.py) -- in one
process, and in two gloo ranks to cover the N>1 path (halo gather + dot all-reduce)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import oracle
from oracle import solvers as S

import host_cases as hc
from oracle_backend import OracleBackend

from legionsolvers_amd import (BiCGStabSolver, CGSolver, Context, GMRESSolver, IndexPartition,
                               IndexSpace, PartitionedVector, Scalar)
from legionsolvers_amd.comm import build_halo_plan
from legionsolvers_amd.index_space import split_equal


# ------------------------------------------------------------------ pure host arithmetic --

def test_equal_partition_matches_oracle_rule():
    from oracle import partition as part
    for n in (1, 4, 20, 100, 101, 1024):
        for p in (1, 3, 4, 8, 10):
            assert split_equal(0, n - 1, p) == part.equal_partition(0, n - 1, p)
    ip = IndexPartition.equal(IndexSpace((0, 0, 0), (3, 4, 4)), 10)
    assert [p.volume for p in ip.pieces] == [0] * 6 + [25] * 4
    assert list(ip.colors_of_rank(1, 2)) == [5, 6, 7, 8, 9]
    with pytest.raises(ValueError):
        ip.colors_of_rank(0, 3)


def test_halo_plan_for_a_linearised_stencil():
    """8 equal row blocks of a 27-point 512^3 grid: each rank reads ny*nz + nz + 1 cells beyond
    either end of its block, from its two neighbours only."""
    n, world, h = 512 ** 3, 8, 512 * 512 + 512 + 1
    owns = split_equal(0, n - 1, world)
    ghosts = [(max(lo - h, 0), min(hi + h, n - 1)) for lo, hi in owns]
    for r in range(world):
        plan = build_halo_plan(r, owns, ghosts)
        assert sorted(t.peer for t in plan.recvs) == [q for q in (r - 1, r + 1) if 0 <= q < world]
        assert sorted(t.peer for t in plan.sends) == [q for q in (r - 1, r + 1) if 0 <= q < world]
        assert all(t.count == h for t in plan.recvs + plan.sends)
        assert (plan.pad_lo, plan.pad_hi) == (h if r else 0, h if r < world - 1 else 0)
    # sends of rank r to q mirror the receives of q from r
    plans = [build_halo_plan(r, owns, ghosts) for r in range(world)]
    for r in range(world):
        for t in plans[r].sends:
            assert any(u.peer == r and u.start == t.start and u.count == t.count for u in plans[t.peer].recvs)


def test_halo_plan_all_to_all_and_empty():
    owns = [(0, 9), (10, 19), (20, 29)]
    full = build_halo_plan(1, owns, [(0, 29)] * 3)       # unstructured matrix: ghost = all of x
    assert [(t.peer, t.start, t.count) for t in full.recvs] == [(0, 0, 10), (2, 20, 10)]
    assert [(t.peer, t.start, t.count) for t in full.sends] == [(0, 10, 10), (2, 10, 10)]
    none = build_halo_plan(0, owns, [(0, -1), (10, 19), (20, 29)])  # a rank with no entries
    assert none.recvs == [] and none.sends == [] and none.pad_lo == none.pad_hi == 0


# ------------------------------------------------------------------ one process ----------

@pytest.fixture()
def ctx():
    return Context(backend=OracleBackend())


@pytest.mark.parametrize("kind", ["csr", "coo"])
def test_cg_single_process_matches_reference_golden(ctx, kind, goldens, krylov_golden):
    prob = goldens["cg_problem"]
    rects, row, col, val = oracle.laplacian_1d(prob["n"])
    pl = hc.mirror_planner(ctx, kind, rects, row, col, val, prob["n"], prob["pieces"], np.float64)
    solver = CGSolver(pl)
    for _ in range(prob["iterations"]):
        solver.step()
    hist = [s.get_value() for s in solver.residual_norm_squared]
    assert hist == krylov_golden["cases"][0]["residual_norm_squared"]
    assert sorted(f"{v:g}" for v in hist) == sorted(goldens["expected_cg_output_sorted_set"])


def test_vector_operations_test02(ctx, goldens):
    """The Test02VectorOperations sequence through PartitionedVector, 1/2/3-D, 10 colours."""
    for dtype in (torch.float32, torch.float64):
        for space in (IndexSpace(0, 99), IndexSpace((0, 0), (9, 9)), IndexSpace((0, 0, 0), (3, 4, 4))):
            ip = IndexPartition.equal(space, 10)
            u, v, w = (PartitionedVector(ctx, n, ip, dtype) for n in "uvw")
            u.constant_fill(1.5)
            v.constant_fill(2.7)
            w.assign(u)
            w.axpy(1.0, v)
            v.xpay(-1.0, u)
            u.axpy(-0.5, v)
            u.axpy(-0.5, w)
            assert f"{u.dot(u).get_value():g}" == goldens["test02_stdout"][0]


def test_scalar_operations_test01(ctx, goldens):
    for dtype in (torch.float32, torch.float64):
        two, ten = Scalar(ctx, 2.0, dtype), Scalar(ctx, 10.0, dtype)
        value = ((two + ten) / (two + two) - two).get_value()
        assert value == goldens["test01_scalar_chain"]["value"]
        assert (-two).get_value() == -2.0 and Scalar(ctx, 4.0, dtype).rsqrt().get_value() == 0.5


@pytest.mark.parametrize("solver_name", ["bicgstab", "gmres"])
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_solver_sequences_match_oracle(ctx, solver_name, dtype):
    """BiCGStab / GMRES(restart) are not pinned by a reference golden; the mirror must issue
    the same operations as the oracle's restatement of the reference headers: identical
    results on one process (same leaf ops, same order)."""
    rects, row, col, val, n = hc.stencil_problem(2, (6, 7), dtype)
    rhs = np.random.default_rng(3).standard_normal(n)
    make = (lambda p: BiCGStabSolver(p)) if solver_name == "bicgstab" else (lambda p: GMRESSolver(p, 4))
    make_o = (lambda p: S.BiCGStabSolver(p)) if solver_name == "bicgstab" else (lambda p: S.GMRESSolver(p, 4))
    opl = hc.oracle_planner("csr", rects, row, col, val, n, 3, dtype, rhs)
    mpl = hc.mirror_planner(ctx, "csr", rects, row, col, val, n, 3, dtype, rhs)
    so, sm = make_o(opl), make(mpl)
    for _ in range(3):
        so.step()
        sm.step()
    np.testing.assert_array_equal(hc.gather_vector(ctx, mpl.get_vector(0, 0)), opl.get_vector(0, 0))
    if solver_name == "bicgstab":
        assert [s.get_value() for s in sm.rho] == [float(v) for v in so.rho]
        assert [s.get_value() for s in sm.omega] == [float(v) for v in so.omega]


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("spaces", [1, 2])
def test_classical_gram_schmidt_gmres_matches_its_oracle(ctx, dtype, spaces):
    """GMRESSolver(orthogonalization="classical"): every inner product of a sweep from the same w
    (SquarePlanner.multi_dot), the updates in one chain (multi_axpy), against the oracle's restatement
    of that variant: identical results on one process (same leaf operations, same order); and not the
    modified sweep's results."""
    rects, row, col, val, n = hc.stencil_problem(2, (6, 7), dtype)
    rhs = np.random.default_rng(3).standard_normal(n)
    opl = hc.oracle_planner("csr", rects, row, col, val, n, 3, dtype, rhs, spaces=spaces)
    mpl = hc.mirror_planner(ctx, "csr", rects, row, col, val, n, 3, dtype, rhs, spaces=spaces)
    so, sm = S.GMRESSolver(opl, 4, "classical"), GMRESSolver(mpl, 4, "classical")
    for _ in range(3):
        so.step()
        sm.step()
    for sp in range(spaces):
        np.testing.assert_array_equal(hc.gather_vector(ctx, mpl.get_vector(0, sp)), opl.get_vector(0, sp))
    got = [[s.get_value() for s in r] for r in sm.inner_products]
    want = [[float(v) for v in r] for r in so.inner_products]
    assert got == want
    opl2 = hc.oracle_planner("csr", rects, row, col, val, n, 3, dtype, rhs, spaces=spaces)
    modified = S.GMRESSolver(opl2, 4)
    for _ in range(3):
        modified.step()
    assert [[float(v) for v in r] for r in modified.inner_products] != want
    with pytest.raises(ValueError):
        GMRESSolver(mpl, 4, "householder")


def test_classical_gram_schmidt_takes_restarts_beyond_one_multi_dot_call(ctx, monkeypatch):
    """A sweep over more basis vectors than one multi_dot / multi_axpy call takes runs in groups: every
    inner product still from the same w, the updates in the same chain, so the same bits as the oracle."""
    rects, row, col, val, n = hc.stencil_problem(2, (6, 7), np.float64)
    rhs = np.random.default_rng(3).standard_normal(n)
    opl = hc.oracle_planner("csr", rects, row, col, val, n, 3, np.float64, rhs)
    mpl = hc.mirror_planner(ctx, "csr", rects, row, col, val, n, 3, np.float64, rhs)
    monkeypatch.setattr(type(ctx.backend), "MULTI_MAX", 3)
    so, sm = S.GMRESSolver(opl, 8, "classical"), GMRESSolver(mpl, 8, "classical")
    for _ in range(2):
        so.step()
        sm.step()
    np.testing.assert_array_equal(hc.gather_vector(ctx, mpl.get_vector(0, 0)), opl.get_vector(0, 0))
    assert [[s.get_value() for s in r] for r in sm.inner_products] == [[float(v) for v in r] for r in so.inner_products]


@pytest.mark.parametrize("kind", ["csr", "coo"])
@pytest.mark.parametrize("pieces", [1, 3])
def test_transposed_product_matches_oracle(ctx, kind, pieces, monkeypatch):
    """SquarePlanner.rmatvec on an unsymmetric matrix against the oracle's entry-by-entry loop and
    against scipy; a second matrix into the same space accumulates; <A x, y> = <x, A^T y>."""
    import scipy.sparse as sp
    rects, row, col, val, n = hc.stencil_problem(4, (5, 4, 3))
    rng = np.random.default_rng(11)
    val = rng.standard_normal(val.size)
    y = rng.standard_normal(n)
    pl = hc.mirror_planner(ctx, kind, rects, row, col, val, n, pieces, np.float64, y)
    pl.allocate_workspace(2)
    pl.rmatvec(2, 1)
    got = pl.get_vector(2, 0).to_host().numpy()
    want = np.zeros(n)
    oracle.csr_rmatvec(rects, 0, col, val, y, 0, want, 0, n - 1)
    np.testing.assert_allclose(got, want, rtol=1e-13, atol=1e-14)
    np.testing.assert_allclose(got, sp.coo_matrix((val, (row, col)), shape=(n, n)).T @ y, rtol=1e-12, atol=1e-13)
    pl.matvec(3, 2)  # <A (A^T y), y> = <A^T y, A^T y>
    lhs, rhs = pl.dot(3, 1).get_value(), pl.dot(2, 2).get_value()
    assert abs(lhs - rhs) <= 1e-12 * abs(rhs)
    # transposed in several runs of entries: the runs' products accumulate to the same sums
    from legionsolvers_amd.matrices import TransposedBlock
    monkeypatch.setattr(TransposedBlock, "CHUNK", 157)
    pl.row_partitioned_matrices[0].matrix._transposed = None
    pl.rmatvec(2, 1)
    assert len(pl.row_partitioned_matrices[0].matrix._transposed.parts) > 3
    np.testing.assert_allclose(pl.get_vector(2, 0).to_host().numpy(), want, rtol=1e-13, atol=1e-13)
    pl.add_row_partitioned_matrix(pl.row_partitioned_matrices[0].matrix, 0, 0)
    pl.rmatvec(2, 1)
    np.testing.assert_allclose(pl.get_vector(2, 0).to_host().numpy(), 2 * want, rtol=1e-13, atol=1e-14)


def test_planner_rejects_mismatched_inputs(ctx):
    ip4 = IndexPartition.equal(IndexSpace(0, 19), 4)
    ip2 = IndexPartition.equal(IndexSpace(0, 19), 2)
    a, b = PartitionedVector(ctx, "a", ip4), PartitionedVector(ctx, "b", ip2)
    with pytest.raises(ValueError):
        a.axpy(1.0, b)
    with pytest.raises(TypeError):
        a.axpy(1.0, PartitionedVector(ctx, "c", ip4, torch.float32))
    with pytest.raises(TypeError):
        a.axpy(1.0, 2.0, 3.0, 4.0, a)
    from legionsolvers_amd import SquarePlanner
    pl = SquarePlanner(ctx)
    pl.add_sol_vector(a)
    with pytest.raises(ValueError):
        pl.add_rhs_vector(b)


# ------------------------------------------------------------------ two gloo ranks ---------

def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ctx = Context(backend=OracleBackend())
        assert (ctx.rank, ctx.world_size) == (rank, world)
        results = {}
        # (a) Test05/06 over 2 ranks x 2 pieces: halo gather between neighbours + all-reduce
        for kind in ("csr", "coo"):
            rects, row, col, val = oracle.laplacian_1d(100)
            pl = hc.mirror_planner(ctx, kind, rects, row, col, val, 100, 4, np.float64)
            assert pl.row_partitioned_matrices[0].plan.recvs, "ranks must exchange ghost cells"
            solver = CGSolver(pl)
            for _ in range(10):
                solver.step()
            results[f"cg_{kind}"] = [s.get_value() for s in solver.residual_norm_squared]
            results[f"cg_{kind}_sol"] = hc.gather_vector(ctx, pl.get_vector(0, 0))
        # (b) two systems, 27-point stencil, BiCGStab: wide halos, batched multi-space dots
        rects, row, col, val, n = hc.stencil_problem(4, (6, 4, 5))
        rhs = np.random.default_rng(5).standard_normal(n)
        pl = hc.mirror_planner(ctx, "csr", rects, row, col, val, n, 2, np.float64, rhs, spaces=2)
        solver = BiCGStabSolver(pl)
        for _ in range(3):
            solver.step()
        results["bicgstab_sol"] = hc.gather_vector(ctx, pl.get_vector(0, 1))
        # (c) GMRES reads SOL, a vector without halo room: staged ghost buffer
        pl = hc.mirror_planner(ctx, "coo", rects, row, col, val, n, 4, np.float64, rhs)
        solver = GMRESSolver(pl, 3)
        solver.step()
        solver.step()
        results["gmres_sol"] = hc.gather_vector(ctx, pl.get_vector(0, 0))
        # (c') the classical Gram-Schmidt variant: k inner products per all-reduce
        pl = hc.mirror_planner(ctx, "csr", rects, row, col, val, n, 4, np.float64, rhs)
        solver = GMRESSolver(pl, 3, "classical")
        solver.step()
        solver.step()
        results["gmres_classical_sol"] = hc.gather_vector(ctx, pl.get_vector(0, 0))
        # (d) transposed product of an unsymmetric matrix: the halo exchange run backwards
        val_u = np.random.default_rng(7).standard_normal(val.size)
        for kind in ("csr", "coo"):
            pl = hc.mirror_planner(ctx, kind, rects, row, col, val_u, n, 4, np.float64, rhs)
            pl.allocate_workspace(1)
            pl.rmatvec(2, 1)
            results[f"rmatvec_{kind}"] = hc.gather_vector(ctx, pl.get_vector(2, 0))
        if rank == 0:
            torch.save(results, os.path.join(out_dir, "results.pt"))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_planner_bets_are_a_function_of_the_call_sequence():
    """The planner's fused leaves are bets on the caller's next operation (a product that leaves the
    shares of the dot that usually follows, an update held back for the dot or the xpay that usually
    follows), kept up while they are won.  Which kernel adds up a dot product therefore depends on the
    calls the planner has seen -- and on nothing else: the same sequence of calls gives the same bits,
    a different history (here: a run of plain products nobody takes a dot of, before the solver starts)
    gives values that agree to rounding, and with the fused leaves off (Context(fuse=False), LSK_FUSE=0)
    there is no history at all."""
    rects, row, col, val, n = hc.stencil_problem(2, (9, 11), np.float64)
    rhs = np.random.default_rng(17).standard_normal(n)

    def run(prelude, fuse):
        c = Context(backend=OracleBackend(), fuse=fuse)
        pl = hc.mirror_planner(c, "csr", rects, row, col, val, n, 3, np.float64, rhs)
        solver = CGSolver(pl)
        for _ in range(prelude):       # Q <- A P: CG's first step overwrites Q with the same product
            pl.matvec(CGSolver.Q, CGSolver.P)
        for _ in range(6):
            solver.step()
        return [s.get_value() for s in solver.residual_norm_squared], pl.get_vector(0, 0).to_host().numpy().copy()

    for prelude in (0, 9):
        a, b = run(prelude, True), run(prelude, True)
        assert a[0] == b[0] and np.array_equal(a[1], b[1])
    plain, after_products = run(0, False), run(9, False)
    assert plain[0] == after_products[0] and np.array_equal(plain[1], after_products[1])
    fused, fused_after_products = run(0, True), run(9, True)
    np.testing.assert_allclose(fused[0], fused_after_products[0], rtol=1e-12)
    np.testing.assert_allclose(fused[0], plain[0], rtol=1e-12)


def _bootstrap_worker(rank, world, port, out_dir):
    """The unique-id bootstrap of the library's RCCL transport over the torch.distributed store: rank 0
    makes the id (ncclGetUniqueId through the C ABI, which needs no GPU), the other reads it."""
    import ctypes
    from torch.distributed.distributed_c10d import _get_default_store
    from legionsolvers_amd import _lib
    from legionsolvers_amd.comm import UNIQUE_ID_BYTES, bootstrap_unique_id
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        made = []

        def make_id():
            buf = ctypes.create_string_buffer(UNIQUE_ID_BYTES)
            _lib.call("lsk_comm_unique_id", buf)
            made.append(buf.raw)
            return buf.raw

        ids = [bootstrap_unique_id(rank, make_id, _get_default_store(), f"test/{k}") for k in range(2)]
        assert len(made) == (2 if rank == 0 else 0)          # only rank 0 asks RCCL for ids
        everyone = [None] * world
        dist.all_gather_object(everyone, ids)
        assert all(e == everyone[0] for e in everyone), "every rank holds rank 0's ids"
        assert ids[0] != ids[1] and all(len(i) == UNIQUE_ID_BYTES for i in ids)
        # a CPU context keeps torch.distributed as its transport (no GPU to bind an RCCL communicator to)
        ctx = Context(backend=OracleBackend())
        assert ctx.comm.transport == "torch" and ctx.comm.rccl_ranks is None
        if rank == 0:
            open(os.path.join(out_dir, "ok"), "w").write("ok")
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_unique_id_bootstrap_two_gloo_ranks(tmp_path):
    mp.spawn(_bootstrap_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    assert os.path.exists(os.path.join(tmp_path, "ok"))


def test_unique_id_bootstrap_rules():
    from legionsolvers_amd.comm import Communicator, UNIQUE_ID_BYTES, bootstrap_unique_id

    class Store(dict):
        def set(self, k, v):
            self[k] = v

        def get(self, k):
            return self[k]

    store = Store()
    uid = bytes(range(UNIQUE_ID_BYTES))
    assert bootstrap_unique_id(0, lambda: uid, store, "k") == uid
    assert bootstrap_unique_id(3, lambda: pytest.fail("only rank 0 makes the id"), store, "k") == uid
    with pytest.raises(ValueError):
        bootstrap_unique_id(0, lambda: b"short", store, "k2")
    store.set("bad", b"x")
    with pytest.raises(RuntimeError):
        bootstrap_unique_id(1, None, store, "bad")
    with pytest.raises(ValueError):
        Communicator(transport="mpi")


def test_bench_starts_its_own_ranks_only_when_nobody_else_did(monkeypatch):
    """`python bench.py --gpus N` without a launcher starts the ranks as a child job; under a launcher
    (WORLD_SIZE set) or for one GPU it runs in place."""
    import importlib.util
    import subprocess
    spec = importlib.util.spec_from_file_location("bench_for_test", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.delenv("LSK_TRANSPORT", raising=False)
    monkeypatch.delenv("LSK_DIST_BACKEND", raising=False)
    calls = []
    monkeypatch.setattr(subprocess, "call", lambda cmd, **kw: calls.append((cmd, kw["env"].get("LSK_TRANSPORT"))) or 7)
    bench.launch_ranks_if_needed(["--steps", "3"])                     # one GPU: in place
    bench.launch_ranks_if_needed(["--gpus", "1"])
    monkeypatch.setenv("WORLD_SIZE", "4")
    bench.launch_ranks_if_needed(["--gpus", "4"])                     # under a launcher: in place
    assert not calls
    monkeypatch.delenv("WORLD_SIZE")
    with pytest.raises(SystemExit) as exc:
        bench.launch_ranks_if_needed(["--gpus", "4", "--steps", "3"])
    assert exc.value.code == 7                                        # the child's code
    assert [t for _, t in calls] == [None, "torch"]                   # second attempt on torch.distributed's transport
    cmd = calls[0][0]
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and cmd[cmd.index("--nproc-per-node") + 1] == "4"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[-4:] == ["--gpus", "4", "--steps", "3"]
    calls.clear()
    monkeypatch.setattr(subprocess, "call", lambda cmd, **kw: calls.append(cmd) or 0)
    with pytest.raises(SystemExit) as exc:
        bench.launch_ranks_if_needed(["--gpus=2"])
    assert exc.value.code == 0 and len(calls) == 1


def test_two_rank_gloo_matches_oracle(tmp_path, krylov_golden):
    mp.spawn(_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    res = torch.load(os.path.join(tmp_path, "results.pt"), weights_only=False)
    golden = krylov_golden["cases"][0]
    for kind in ("csr", "coo"):
        np.testing.assert_allclose(res[f"cg_{kind}"], golden["residual_norm_squared"], rtol=1e-12)
        np.testing.assert_allclose(res[f"cg_{kind}_sol"], golden["x_final"], rtol=1e-10)
    rects, row, col, val, n = hc.stencil_problem(4, (6, 4, 5))
    rhs = np.random.default_rng(5).standard_normal(n)
    opl = hc.oracle_planner("csr", rects, row, col, val, n, 2, np.float64, rhs, spaces=2)
    so = S.BiCGStabSolver(opl)
    for _ in range(3):
        so.step()
    np.testing.assert_allclose(res["bicgstab_sol"], opl.get_vector(0, 1), rtol=1e-9, atol=1e-12)
    opl = hc.oracle_planner("coo", rects, row, col, val, n, 4, np.float64, rhs)
    so = S.GMRESSolver(opl, 3)
    so.step()
    so.step()
    np.testing.assert_allclose(res["gmres_sol"], opl.get_vector(0, 0), rtol=1e-9, atol=1e-12)
    opl = hc.oracle_planner("csr", rects, row, col, val, n, 4, np.float64, rhs)
    so = S.GMRESSolver(opl, 3, "classical")
    so.step()
    so.step()
    np.testing.assert_allclose(res["gmres_classical_sol"], opl.get_vector(0, 0), rtol=1e-9, atol=1e-12)
    val_u = np.random.default_rng(7).standard_normal(val.size)
    want = np.zeros(n)
    oracle.csr_rmatvec(rects, 0, col, val_u, rhs, 0, want, 0, n - 1)
    for kind in ("csr", "coo"):
        np.testing.assert_allclose(res[f"rmatvec_{kind}"], want, rtol=1e-12, atol=1e-13)


# ------------------------------------------------------------------ eight gloo ranks -------

def _worker8(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ctx = Context(backend=OracleBackend())
        results = {}
        # BenchmarkStencil's layout, one row block per rank.  (32, 8, 8): four grid planes per
        # rank, halos from the two neighbours, the two inner planes need none; (4, 8, 8): half a plane per rank, so the halo
        # (up to 73 cells) reaches three ranks away on each side.
        for extent in ((32, 8, 8), (4, 8, 8)):
            rects, row, col, val, n = hc.stencil_problem(4, extent)
            pl = hc.mirror_planner(ctx, "csr", rects, row, col, val, n, world, np.float64)
            plan = pl.row_partitioned_matrices[0].plan
            per = n // world
            mine = col[rects[rank * per, 0]:rects[(rank + 1) * per - 1, 1] + 1]
            want_peers = [q for q in range(int(mine.min()) // per, int(mine.max()) // per + 1) if q != rank]
            if extent[0] == 32:
                assert want_peers == [q for q in (rank - 1, rank + 1) if 0 <= q < world]
            elif 2 <= rank <= 5:
                assert max(abs(q - rank) for q in want_peers) >= 2
            split = pl.row_partitioned_matrices[0].split
            if extent[0] == 32:   # rows that read no ghost cell run ahead of the halo
                runs = split[0]
                assert [r[0] for r in runs] == ([0, 1] if rank == 0 else [1, 0] if rank == world - 1 else [1, 0, 1])
                assert sum(r[2] - r[1] for r in runs) == per
            else:                 # half a plane per rank: every row needs the halo
                assert split is None
            assert sorted(t.peer for t in plan.recvs) == want_peers, (rank, plan.recvs)
            assert sorted(t.peer for t in plan.sends) == want_peers, (rank, plan.sends)
            solver = CGSolver(pl)
            for _ in range(12):
                solver.step()
            results[f"cg_{extent[0]}"] = [s.get_value() for s in solver.residual_norm_squared]
            results[f"sol_{extent[0]}"] = hc.gather_vector(ctx, pl.get_vector(0, 0))
        if rank == 0:
            torch.save(results, os.path.join(out_dir, "results8.pt"))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_eight_rank_gloo_matches_oracle(tmp_path):
    """The 8-rank job's host logic (one row block per rank, halo plan, halo placement, dot
    all-reduce) on CPU tensors, against the oracle with 8 pieces."""
    mp.spawn(_worker8, args=(8, _free_port(), str(tmp_path)), nprocs=8, join=True)
    res = torch.load(os.path.join(tmp_path, "results8.pt"), weights_only=False)
    for extent in ((32, 8, 8), (4, 8, 8)):
        rects, row, col, val, n = hc.stencil_problem(4, extent)
        opl = hc.oracle_planner("csr", rects, row, col, val, n, 8, np.float64)
        so = S.CGSolver(opl)
        for _ in range(12):
            so.step()
        np.testing.assert_allclose(res[f"cg_{extent[0]}"], [float(v) for v in so.residual_norm_squared], rtol=1e-12)
        np.testing.assert_allclose(res[f"sol_{extent[0]}"], opl.get_vector(0, 0), rtol=1e-10, atol=1e-13)


def test_planner_compresses_the_structure_and_falls_back(monkeypatch):
    """Host logic of the set-up-time 32-bit structure on the oracle backend: made for back-to-back
    rects (and used: the product then gets int32 arrays), not made for gapped rects or with
    LSK_COMPRESS=0; the CG history is the same in all three cases that share a matrix."""
    rects, row, col, val, n = hc.stencil_problem(3, (6, 5, 7), np.float64)
    seen = []

    class Spy(OracleBackend):
        def csr_spmv(self, rows, nnz, rects, k_base, col, *rest, **kw):
            seen.append((rects.dtype, col.dtype))
            return super().csr_spmv(rows, nnz, rects, k_base, col, *rest, **kw)

    histories = {}
    for label, env, rc in (("compressed", "1", rects), ("off", "0", rects)):
        monkeypatch.setenv("LSK_COMPRESS", env)
        seen.clear()
        ctx = Context(backend=Spy())
        mpl = hc.mirror_planner(ctx, "csr", rc, row, col, val, n, 2, np.float64)
        matrix = mpl.row_partitioned_matrices[0].matrix
        assert (matrix.col32 is not None) == (label == "compressed")
        solver = CGSolver(mpl)
        for _ in range(4):
            solver.step()
        histories[label] = [s.get_value() for s in solver.residual_norm_squared]
        want = (torch.int32, torch.int32) if label == "compressed" else (torch.int64, torch.int64)
        assert seen and all(pair == want for pair in seen)
    assert histories["compressed"] == histories["off"]
    monkeypatch.setenv("LSK_COMPRESS", "1")
    gapped = rects.copy()
    gapped[gapped[:, 1] > gapped[:, 0], 1] -= 1
    ctx = Context(backend=Spy())
    mpl = hc.mirror_planner(ctx, "csr", gapped, row, col, val, n, 2, np.float64)
    matrix = mpl.row_partitioned_matrices[0].matrix
    assert matrix.col32 is None and not matrix.rowoff


def test_held_back_axpy_is_invisible_to_readers():
    """axpy(defer=True) on the oracle backend: reading either vector makes the update happen
    first; a following xpay on the source fuses with it when nothing looked in between; both
    ways give the values of the plain sequence."""
    from legionsolvers_amd import Scalar
    ctx = Context(backend=OracleBackend())
    part = IndexPartition.equal(IndexSpace(0, 99), 3)
    rng = np.random.default_rng(5)
    init = {k: torch.from_numpy(rng.standard_normal(100)) for k in "zpr"}
    a, b = Scalar(ctx, 0.75, torch.float64), Scalar(ctx, -1.25, torch.float64)

    def fresh():
        out = {}
        for k in "zpr":
            v = PartitionedVector(ctx, k, part, torch.float64)
            v.owned.copy_(init[k])
            out[k] = v
        return out

    plain = fresh()
    plain["z"].axpy(a, plain["p"])
    plain["p"].xpay(b, plain["r"])
    outcomes = []
    fused = fresh()
    fused["z"].axpy(a, fused["p"], defer=True, on_outcome=outcomes.append)
    assert fused["z"]._deferred is not None and fused["p"]._deferred is fused["z"]._deferred
    fused["p"].xpay(b, fused["r"])
    assert outcomes == [True] and fused["z"]._deferred is None
    peeked = fresh()
    peeked["z"].axpy(a, peeked["p"], defer=True, on_outcome=outcomes.append)
    seen = peeked["z"].to_host().clone()          # a reader in between: the update runs now
    assert outcomes == [True, False] and torch.equal(seen, plain["z"].to_host())
    peeked["p"].xpay(b, peeked["r"])
    for k in "zpr":
        assert torch.equal(fused[k].to_host(), plain[k].to_host())
        assert torch.equal(peeked[k].to_host(), plain[k].to_host())
    # writing the source first also settles the update before the write
    w = fresh()
    w["z"].axpy(a, w["p"], defer=True)
    w["p"].scal(b)
    assert torch.equal(w["z"].to_host(), plain["z"].to_host())


def test_held_back_axpy_fuses_with_the_next_dot():
    """axpy(defer=True) followed by a dot product with the updated vector: one fused leaf, same
    values as the plain sequence (oracle backend); with a third vector, with itself, with the
    source; a dot product of two other vectors leaves the update pending."""
    from legionsolvers_amd import Scalar
    ctx = Context(backend=OracleBackend())
    part = IndexPartition.equal(IndexSpace(0, 199), 4)
    rng = np.random.default_rng(6)
    init = {k: torch.from_numpy(rng.standard_normal(200)) for k in "zpwu"}
    a = Scalar(ctx, -0.375, torch.float64)

    def fresh():
        out = {}
        for k in "zpwu":
            v = PartitionedVector(ctx, k, part, torch.float64)
            v.owned.copy_(init[k])
            out[k] = v
        return out

    for partner in "wzp":
        plain, held = fresh(), fresh()
        plain["z"].axpy(a, plain["p"])
        want = plain["z"].dot(plain[partner]).get_value()
        outcomes = []
        held["z"].axpy(a, held["p"], defer=True, on_outcome=outcomes.append)
        assert held["w"].dot(held["u"]).get_value() == plain["w"].dot(plain["u"]).get_value()
        assert held["z"]._deferred is not None          # untouched by a dot of two other vectors
        got = (held[partner].dot(held["z"]) if partner == "w" else held["z"].dot(held[partner])).get_value()
        assert outcomes == [True] and held["z"]._deferred is None and held["p"]._deferred is None
        assert abs(got - want) <= 1e-13 * abs(want)
        assert torch.equal(held["z"].to_host(), plain["z"].to_host())


def test_workspace_place_is_chosen_by_gain_not_by_minimum(ctx):
    """SquarePlanner places its workspace by the product's time behind 0, 1 and 2 blocks of a third of the device:
    a deeper place is taken only when it beats the place chosen so far by the minimum gain, so that timer noise
    never moves the vectors; and on the CPU (and with LSK_PLACE_VECTORS=0) nothing is timed at all."""
    from legionsolvers_amd.square_planner import SquarePlanner
    choose, gain = SquarePlanner._choose_place, SquarePlanner.PLACE_MIN_GAIN
    assert choose([4.76], gain) == 0
    assert choose([4.76, 4.43, 4.42], gain) == 1       # the second block gains nothing over the first
    assert choose([4.58, 4.60, 4.64], gain) == 0       # all alike
    assert choose([4.58, 4.54, 4.45], gain) == 2       # only the deepest pays (against the nearest: 2.8 %)
    assert choose([4.58, 4.57, 4.53], gain) == 0       # 1.1 % in all: stays
    assert choose([4.76, 4.43, 4.30], gain) == 2       # each step pays
    assert choose([], gain) == 0
    rects, row, col, val, n = hc.stencil_problem(1, (100,), np.float64)
    mpl = hc.mirror_planner(ctx, "csr", rects, row, col, val, n, 2, np.float64, np.ones(n))
    assert mpl._room_in_front_of_the_workspace(3 * n * 8) == [] and mpl.workspace_placement is None
    CGSolver(mpl)
    assert mpl.workspace_placement is None and len(mpl.workspace_vectors) == 3
