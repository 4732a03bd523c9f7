"""A solver step recorded as a HIP graph (solvers.CapturedStep): every replay against the oracle's
step of the same number and against a twin solver stepping eagerly, with eager work and
synchronisations between the replays."""
import numpy as np
import pytest
import torch

from oracle import solvers as S

import host_cases as hc

from legionsolvers_amd import BiCGStabSolver, CGSolver, GMRESSolver

pytestmark = pytest.mark.gpu


def _make(solver_name, planner, oracle_side=False):
    mod = S if oracle_side else None
    if solver_name == "cg":
        return (mod.CGSolver if mod else CGSolver)(planner)
    if solver_name == "bicgstab":
        return (mod.BiCGStabSolver if mod else BiCGStabSolver)(planner)
    return (mod.GMRESSolver if mod else GMRESSolver)(planner, 4)


def _scalars(solver_name, solver):
    if solver_name == "cg":
        return [solver.residual_norm_squared[-1]]
    if solver_name == "bicgstab":
        return [solver.rho[-1], solver.alpha[-1], solver.omega[-1]]
    return [solver.inner_products[j + 1][j] for j in range(solver.restart)]


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("kind,pieces,spaces", [("csr", 1, 1), ("csr", 3, 2), ("coo", 2, 1)])
@pytest.mark.parametrize("solver_name", ["cg", "bicgstab", "gmres"])
def test_replayed_steps_follow_the_oracle_and_the_eager_solver(gpu_ctx, solver_name, kind, pieces, spaces, dtype):
    print(f"case recorded step {solver_name} {kind} pieces={pieces} spaces={spaces} {np.dtype(dtype).name}", flush=True)
    rects, row, col, val, n = hc.stencil_problem(4, (7, 9, 11), dtype)
    rhs = np.random.default_rng(77).standard_normal(n)
    if solver_name != "cg":
        spaces = 1
    opl = hc.oracle_planner(kind, rects, row, col, val, n, pieces, dtype, rhs, spaces)
    so = _make(solver_name, opl, oracle_side=True)
    eager = _make(solver_name, hc.mirror_planner(gpu_ctx, kind, rects, row, col, val, n, pieces, dtype, rhs, spaces))
    mpl = hc.mirror_planner(gpu_ctx, kind, rects, row, col, val, n, pieces, dtype, rhs, spaces)
    sm = _make(solver_name, mpl)
    warm, replays = 2, 4
    for _ in range(warm):
        so.step()
        eager.step()
        sm.step()
    recorded = sm.capture()
    entries = len(sm.residual_norm_squared) if solver_name == "cg" else None
    rtol = 1e-10 if dtype == np.float64 else 2e-3
    for i in range(replays):
        so.step()
        eager.step()
        recorded.replay()
        torch.cuda.synchronize()
        got = [s.get_value() for s in _scalars(solver_name, sm)]
        np.testing.assert_allclose(got, [float(v) for v in _scalars(solver_name, so)], rtol=rtol,
                                   err_msg=f"replay {i} against the oracle")
        np.testing.assert_allclose(got, [s.get_value() for s in _scalars(solver_name, eager)],
                                   rtol=1e-13 if dtype == np.float64 else 1e-5, err_msg=f"replay {i} against eager steps")
        for space in range(spaces):
            want = eager.planner.get_vector(0, space).to_host().numpy().astype(np.float64)
            sol = mpl.get_vector(0, space).to_host().numpy().astype(np.float64)
            assert np.abs(sol - want).max() <= (1e-12 if dtype == np.float64 else 1e-4) * max(np.abs(want).max(), 1e-300)
        # eager work of the program's own between the replays
        scratch = torch.randn(4096, device="cuda")
        assert torch.isfinite(scratch.sum()).item()
    if entries is not None:
        assert len(sm.residual_norm_squared) == entries   # the history of replayed steps is not kept
    assert recorded.replays == replays
    # and eager steps go on from where the replays stopped
    so.step()
    sm.step()
    got = [s.get_value() for s in _scalars(solver_name, sm)]
    np.testing.assert_allclose(got, [float(v) for v in _scalars(solver_name, so)], rtol=rtol)


@pytest.mark.parametrize("solver_name", ["cg", "bicgstab"])
def test_a_recording_of_several_steps(gpu_ctx, solver_name):
    """capture(steps=3): every replay is three steps; the carried scalars come from the LAST of them."""
    rects, row, col, val, n = hc.stencil_problem(4, (6, 7, 8), np.float64)
    rhs = np.random.default_rng(78).standard_normal(n)
    so = _make(solver_name, hc.oracle_planner("csr", rects, row, col, val, n, 2, np.float64, rhs, 1), oracle_side=True)
    sm = _make(solver_name, hc.mirror_planner(gpu_ctx, "csr", rects, row, col, val, n, 2, np.float64, rhs, 1))
    for _ in range(2):
        so.step()
        sm.step()
    recorded = sm.capture(steps=3)
    for i in range(3):
        for _ in range(3):
            so.step()
        recorded.replay()
        torch.cuda.synchronize()
        np.testing.assert_allclose([s.get_value() for s in _scalars(solver_name, sm)],
                                   [float(v) for v in _scalars(solver_name, so)], rtol=1e-9, err_msg=f"replay {i}")
    with pytest.raises(ValueError):
        sm.capture(steps=0)


def test_recorded_steps_through_the_binned_coo_copy(gpu_ctx, monkeypatch):
    """BiCGStab on an unstructured COO matrix whose products go through the binned copy (two passes through
    LDS, a kernel pair with a raised LDS limit): recorded after two eager steps, replays follow the oracle."""
    from legionsolvers_amd import COOMatrix
    monkeypatch.setenv("LSK_COO_BLOCK", "1")
    monkeypatch.setenv("LSK_COO_BINNED", "2")
    monkeypatch.setattr(COOMatrix, "BLOCK_MIN_WINDOW", 0)
    monkeypatch.setattr(COOMatrix, "BINNED_BLOCK_COLS", 256)
    monkeypatch.setattr(COOMatrix, "BINNED_TILE_ROWS", 512)
    rng = np.random.default_rng(39)
    n, per = 6000, 12
    row = np.repeat(np.arange(n), per).astype(np.int64)
    col = rng.integers(0, n, size=n * per).astype(np.int64)
    col[::per] = np.arange(n)
    val = rng.uniform(-1, 1, size=n * per)
    val[::per] = 2.0 * per
    rhs = rng.standard_normal(n)
    so = S.BiCGStabSolver(hc.oracle_planner("coo", None, row, col, val, n, 1, np.float64, rhs))
    mpl = hc.mirror_planner(gpu_ctx, "coo", None, row, col, val, n, 1, np.float64, rhs)
    assert mpl.row_partitioned_matrices[0].matrix.binned is not None
    sm = BiCGStabSolver(mpl)
    for _ in range(2):
        so.step()
        sm.step()
    recorded = sm.capture()
    for i in range(4):
        so.step()
        recorded.replay()
        torch.cuda.synchronize()
        np.testing.assert_allclose([s.get_value() for s in _scalars("bicgstab", sm)],
                                   [float(v) for v in _scalars("bicgstab", so)], rtol=1e-9, err_msg=f"replay {i}")
    want = so.planner.get_vector(0, 0)
    got = mpl.get_vector(0, 0).to_host().numpy()
    assert np.abs(got - want).max() <= 1e-9 * np.abs(want).max()


@pytest.mark.parametrize("sliced,compress", [("0", "1"), ("0", "0")])
def test_recorded_steps_on_the_general_csr_kernels(gpu_ctx, monkeypatch, sliced, compress):
    """CG with the sliced-pattern copy off: the products of the recorded step run on the 32-bit copy of
    the structure, or on the int64 arrays as handed over."""
    monkeypatch.setenv("LSK_SLICED", sliced)
    monkeypatch.setenv("LSK_COMPRESS", compress)
    rects, row, col, val, n = hc.stencil_problem(4, (9, 8, 7), np.float64)
    rhs = np.random.default_rng(79).standard_normal(n)
    so = S.CGSolver(hc.oracle_planner("csr", rects, row, col, val, n, 3, np.float64, rhs))
    mpl = hc.mirror_planner(gpu_ctx, "csr", rects, row, col, val, n, 3, np.float64, rhs)
    matrix = mpl.row_partitioned_matrices[0].matrix
    assert not matrix.sliced and (matrix.col32 is not None) == (compress == "1")
    sm = CGSolver(mpl)
    for _ in range(2):
        so.step()
        sm.step()
    recorded = sm.capture()
    for i in range(4):
        so.step()
        recorded.replay()
        torch.cuda.synchronize()
        np.testing.assert_allclose(sm.residual_norm_squared[-1].get_value(), float(so.residual_norm_squared[-1]),
                                   rtol=1e-10, err_msg=f"replay {i}")


def test_recording_is_refused_where_it_cannot_work(gpu_ctx, monkeypatch):
    rects, row, col, val, n = hc.stencil_problem(4, (5, 5, 5), np.float64)
    mpl = hc.mirror_planner(gpu_ctx, "csr", rects, row, col, val, n, 1, np.float64, np.ones(n), 1)
    solver = CGSolver(mpl)
    monkeypatch.setattr(gpu_ctx.comm, "world_size", 2)
    with pytest.raises(NotImplementedError):
        solver.capture()


def test_c_abi_trace_records_and_replays(gpu_ctx):
    """lsk_stream_begin_capture / lsk_stream_end_capture / lsk_trace_replay through ctypes on a stream of
    the library's own: recording executes nothing, every replay is the recorded leaves once more, with
    eager work and synchronisations between the replays; misuse is refused."""
    import ctypes
    from legionsolvers_amd import _lib
    P = lambda t: ctypes.c_void_p(t.data_ptr())  # noqa: E731
    stream = ctypes.c_void_p(0)
    _lib.call("lsk_stream_create", ctypes.byref(stream))
    n = 100003
    x = torch.arange(n, dtype=torch.float64, device="cuda") / n
    y = torch.zeros(n, dtype=torch.float64, device="cuda")
    alpha = torch.tensor([0.5], dtype=torch.float64, device="cuda")
    out = torch.zeros(1, dtype=torch.float64, device="cuda")
    ws = torch.empty(int(_lib.lib().lsk_dot_workspace_bytes(0)), dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()

    def leaves():
        # y <- y + alpha x, then out <- y . x
        _lib.call("lsk_axpy_dev_f64", n, 1, P(alpha), None, None, None, P(x), P(y), stream)
        _lib.call("lsk_dot_f64", n, P(y), P(x), P(out), P(ws), ws.numel(), stream)

    def recording():
        flag = ctypes.c_int(-1)
        _lib.call("lsk_stream_is_recording", stream, ctypes.byref(flag))
        return flag.value

    assert recording() == 0
    leaves()                                   # once eagerly: first launches are not recordable
    _lib.call("lsk_stream_synchronize", stream)
    y.zero_()
    torch.cuda.synchronize()
    trace = ctypes.c_void_p(0)
    _lib.call("lsk_stream_begin_capture", stream)
    with pytest.raises(_lib.LskError):          # (the stream is recording already)
        _lib.call("lsk_stream_begin_capture", stream)
    assert recording() == 1
    # calls that wait for the device are refused before they enqueue anything: the recording stays whole
    host = ctypes.c_double(-1.0)
    for name, args in (("lsk_stream_synchronize", (stream,)), ("lsk_stream_check", (stream,)),
                       ("lsk_dot_host_f64", (n, P(y), P(x), ctypes.byref(host), stream))):
        with pytest.raises(_lib.LskError, match="recording"):
            _lib.call(name, *args)
    leaves()
    _lib.call("lsk_stream_end_capture", stream, ctypes.byref(trace))
    assert trace.value and recording() == 0
    _lib.call("lsk_stream_synchronize", stream)
    assert float(y.abs().max()) == 0.0          # recording executed nothing
    info = (ctypes.c_int64 * 2)()
    _lib.call("lsk_trace_info", trace, info)
    assert info[0] >= 2 and info[1] == 0
    xx = float((x.double() * x.double()).sum())
    for k in range(1, 5):
        _lib.call("lsk_trace_replay", trace, stream)
        _lib.call("lsk_stream_synchronize", stream)
        assert torch.equal(y, (0.5 * k) * x) or float((y - 0.5 * k * x).abs().max()) <= 1e-15
        assert abs(float(out) - 0.5 * k * xx) <= 1e-12 * xx
        assert torch.isfinite(torch.randn(1000, device="cuda").sum()).item()   # eager work between the replays
    _lib.call("lsk_trace_info", trace, info)
    assert info[1] == 4
    _lib.call("lsk_trace_destroy", trace)
    # ending a recording that was never begun is an error, not a crash
    other = ctypes.c_void_p(0)
    with pytest.raises(_lib.LskError):
        _lib.call("lsk_stream_end_capture", stream, ctypes.byref(other))
    assert not other.value
    with pytest.raises(_lib.LskError):
        _lib.call("lsk_trace_replay", None, stream)
    _lib.call("lsk_stream_destroy", stream)


def test_debug_sync_checks_every_launch_and_refuses_recordings(gpu_ctx):
    """Option debug.sync (LSK_DEBUG_SYNC=1): every launch is followed by a device synchronisation and an error
    check (the reference's debug builds do it at task exit, src/CUDAUtilities.cpp:110-122).  Same results; a
    recording is refused while it is on."""
    import ctypes
    from legionsolvers_amd import _lib
    rects, row, col, val, n = hc.stencil_problem(4, (6, 7, 8), np.float64)
    rhs = np.random.default_rng(80).standard_normal(n)
    histories = []
    for on in (0, 1):
        _lib.call("lsk_set_option", b"debug.sync", on)
        try:
            got = ctypes.c_int(-1)
            _lib.call("lsk_get_option", b"debug.sync", ctypes.byref(got))
            assert got.value == on
            sm = CGSolver(hc.mirror_planner(gpu_ctx, "csr", rects, row, col, val, n, 2, np.float64, rhs, 1))
            for _ in range(5):
                sm.step()
            histories.append([s.get_value() for s in sm.residual_norm_squared])
            if on:
                with pytest.raises(RuntimeError):
                    sm.capture()
                stream = ctypes.c_void_p(0)
                _lib.call("lsk_stream_create", ctypes.byref(stream))
                with pytest.raises(_lib.LskError):
                    _lib.call("lsk_stream_begin_capture", stream)
                _lib.call("lsk_stream_destroy", stream)
        finally:
            _lib.call("lsk_set_option", b"debug.sync", 0)
    assert histories[0] == histories[1]
