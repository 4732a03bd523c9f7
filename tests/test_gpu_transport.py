"""The library's own RCCL transport (comm.LskTransport over lsk_comm_*, lsk_allreduce_sum_*,
lsk_halo_exchange_*) on the one GPU of the test machine: a one-rank communicator with option
comm.skip_single_rank = 0, so that every all-reduce and every transfer is really enqueued through RCCL
-- the all-reduces on the current stream, the transfers on the transport's second stream behind events.
What one GPU cannot show is a second rank; the plan arithmetic for that runs in the gloo and
thread-rank tests."""
import numpy as np
import pytest
import torch

import host_cases as hc
from oracle import solvers as S

from legionsolvers_amd import BiCGStabSolver, CGSolver, Context, GMRESSolver, _lib
from legionsolvers_amd.comm import Communicator

pytestmark = pytest.mark.gpu


@pytest.fixture()
def rccl_on_one_rank():
    _lib.call("lsk_set_option", b"comm.skip_single_rank", 0)
    yield
    _lib.call("lsk_set_option", b"comm.skip_single_rank", 1)


def lsk_context():
    ctx = Context(torch.device("cuda", 0), comm=Communicator(transport="lsk"))
    assert ctx.comm.transport == "lsk" and ctx.comm.rccl_ranks == 1 and ctx.comm.lsk.rccl_rank == 0
    return ctx


@pytest.mark.parametrize("solver_name,kind,pieces", [("cg", "csr", 4), ("bicgstab", "coo", 2), ("gmres", "csr", 3),
                                                     ("gmres_classical", "csr", 2)])
def test_planner_on_the_lsk_transport_follows_the_oracle(rccl_on_one_rank, solver_name, kind, pieces):
    """Every dot product of the solvers crosses ncclAllReduce (one rank, really enqueued) between the
    kernel that leaves it and the kernel that reads it; histories and solutions against the oracle."""
    print(f"case lsk transport {solver_name} {kind} pieces={pieces}", flush=True)
    ctx = lsk_context()
    try:
        rects, row, col, val, n = hc.stencil_problem(4, (12, 10, 14), np.float64)
        rhs = np.random.default_rng(41).standard_normal(n)
        spaces = 2 if solver_name == "cg" else 1
        opl = hc.oracle_planner(kind, rects, row, col, val, n, pieces, np.float64, rhs, spaces)
        mpl = hc.mirror_planner(ctx, kind, rects, row, col, val, n, pieces, np.float64, rhs, spaces)
        if solver_name == "cg":
            so, sm, steps = S.CGSolver(opl), CGSolver(mpl), 8
        elif solver_name == "bicgstab":
            so, sm, steps = S.BiCGStabSolver(opl), BiCGStabSolver(mpl), 5
        elif solver_name == "gmres":
            so, sm, steps = S.GMRESSolver(opl, 5), GMRESSolver(mpl, 5), 2
        else:
            so, sm, steps = S.GMRESSolver(opl, 5, "classical"), GMRESSolver(mpl, 5, "classical"), 2
        for _ in range(steps):
            so.step()
            sm.step()
        if solver_name == "cg":
            np.testing.assert_allclose([s.get_value() for s in sm.residual_norm_squared],
                                       [float(v) for v in so.residual_norm_squared], rtol=1e-10)
        want = opl.get_vector(0, spaces - 1)
        got = mpl.get_vector(0, spaces - 1).to_host().numpy()
        assert np.abs(got - want).max() <= 1e-9 * np.abs(want).max()
    finally:
        ctx.synchronize()
        ctx.comm.close()


@pytest.mark.parametrize("through_rccl", [True, False])
def test_transfers_run_on_the_second_stream_in_order(through_rccl):
    """exchange_start posts the transfers behind what the current stream holds, kernels enqueued
    between start and finish overlap with them, exchange_finish makes the current stream wait: the data
    that arrives is what the current stream had written before start, every round.  The peer is the rank
    itself (ncclSend + ncclRecv of one rank in one group), all a one-GPU machine offers."""
    _lib.call("lsk_set_option", b"comm.skip_single_rank", 0 if through_rccl else 1)
    ctx = lsk_context()
    try:
        be = ctx.backend
        n = 262_657                                   # one halo of the 27-point 512^3 problem on 8 ranks
        for dtype in (torch.float64, torch.float32):
            src, dst = be.empty(2 * n, dtype), be.empty(2 * n, dtype)
            big_a, big_b = be.empty(1 << 24, dtype), be.empty(1 << 24, dtype)
            be.fill(big_a, 1.0)
            be.fill(dst, -1.0)
            for round_ in range(6):
                print(f"case lsk transfer through_rccl={through_rccl} {dtype} round {round_}", flush=True)
                be.fill(src, float(round_))                                     # current stream writes ...
                handle = ctx.comm.exchange_start([(0, src[:n]), (0, src[n:])], [(0, dst[n:]), (0, dst[:n])])
                be.copy(big_b, big_a)                                           # ... overlaps with the transfer ...
                ctx.comm.exchange_finish(handle)
                be.scal(dst, [be.new_scalar(2.0, dtype)])                       # ... and reads what arrived
                ctx.synchronize()
                assert torch.all(dst == 2.0 * round_).item(), round_
                assert torch.all(big_b == 1.0).item()
        # an all-reduce of k scalars between the kernel that leaves them and the one that reads them
        v = torch.arange(1, 65, dtype=torch.float64, device=ctx.device)
        out = be.empty(3, torch.float64)
        for j in range(3):
            be.dot(v, v, out[j:j + 1])
        ctx.comm.all_reduce_sum(out)
        ctx.synchronize()
        assert out.tolist() == [float(sum(i * i for i in range(1, 65)))] * 3
    finally:
        ctx.synchronize()
        ctx.comm.close()
        _lib.call("lsk_set_option", b"comm.skip_single_rank", 1)


def test_the_asked_for_transport_is_reported():
    ctx = Context(torch.device("cuda", 0), comm=Communicator(transport="torch"))
    assert ctx.comm.transport == "torch" and ctx.comm.rccl_ranks is None and ctx.comm.lsk is None
    ctx = Context(torch.device("cuda", 0))           # one rank, nothing asked for: nothing to move, no communicator
    assert ctx.comm.transport == "torch" and ctx.comm.lsk is None
