"""Worker of test_gpu_two_ranks.py: one rank of a 2-rank (or WORLD_SIZE-rank) job whose ranks share cuda:0 and talk
over gloo with host staging.  Runs CG (CSR, two coupled spaces), BiCGStab (COO) and GMRES (CSR)
on the HIP backend and compares every rank's view with the CPU oracle."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)

import host_cases as hc  # noqa: E402
from oracle import solvers as S  # noqa: E402

from legionsolvers_amd import BiCGStabSolver, CGSolver, Context, GMRESSolver  # noqa: E402


def main():
    torch.cuda.set_device(0)
    dist.init_process_group("gloo")
    standin = os.environ.get("LSK_TEST_STANDIN_RCCL") == "1"
    if standin:   # the library's own transport above a stand-in for its RCCL entry points (standin/standin_rccl.py)
        sys.path.insert(0, os.path.join(HERE, "standin"))
        import standin_rccl
        standin_rccl.install()
    ctx = Context(torch.device("cuda", 0))
    world = int(os.environ["WORLD_SIZE"])
    assert ctx.world_size == world and ctx.comm.backend == "gloo"
    hangs = os.environ.get("LSK_TEST_STANDIN_INIT_HANGS") == "1"
    if standin and hangs:   # the communicator never came up: every rank went on with torch.distributed, and says why
        assert ctx.comm.transport == "torch" and ctx.comm.rccl_ranks is None
        assert "was not up after" in ctx.comm.transport_note, ctx.comm.transport_note
        standin = False
    elif standin:
        assert ctx.comm.transport == "lsk" and ctx.comm.rccl_ranks == world and ctx.comm.transport_note is None
    else:
        assert ctx.comm.transport == "torch"
    for solver_name, kind, pieces in (("cg", "csr", 2 * world), ("bicgstab", "coo", world), ("gmres", "csr", 3 * world),
                                      ("cg", "coo", world)):
        print(f"rank {ctx.rank} case {solver_name} {kind} pieces={pieces}", flush=True)
        rects, row, col, val, n = hc.stencil_problem(4, (12, 10, 14), np.float64)
        rhs = np.random.default_rng(41).standard_normal(n)
        spaces = 2 if solver_name == "cg" else 1
        opl = hc.oracle_planner(kind, rects, row, col, val, n, pieces, np.float64, rhs, spaces)
        mpl = hc.mirror_planner(ctx, kind, rects, row, col, val, n, pieces, np.float64, rhs, spaces)
        if kind == "csr" and world == 2:  # the halo overlap path is the one that runs, on the sliced-pattern copy
            assert mpl.row_partitioned_matrices[0].split is not None
            assert mpl.row_partitioned_matrices[0].matrix.sliced
        if solver_name == "cg":
            so, sm, steps = S.CGSolver(opl), CGSolver(mpl), 8
        elif solver_name == "bicgstab":
            so, sm, steps = S.BiCGStabSolver(opl), BiCGStabSolver(mpl), 5
        else:
            so, sm, steps = S.GMRESSolver(opl, 5), GMRESSolver(mpl, 5), 2
        for _ in range(steps):
            so.step()
            sm.step()
        if solver_name == "cg" and kind == "csr" and world == 2:  # both phases of the split product left the shares of p.Ap
            assert mpl._fused_dot_used >= steps - 1, (mpl._fused_dot_issued, mpl._fused_dot_used)
        if solver_name == "cg":
            np.testing.assert_allclose([s.get_value() for s in sm.residual_norm_squared],
                                       [float(v) for v in so.residual_norm_squared], rtol=1e-10)
        want = opl.get_vector(0, spaces - 1)
        got = hc.gather_vector(ctx, mpl.get_vector(0, spaces - 1))
        assert got.shape == want.shape
        assert np.abs(got - want).max() <= 1e-9 * np.abs(want).max(), (solver_name, kind)
    if standin:
        done = standin_rccl.calls()
        assert done["allreduce"] > 50 and done["exchange"] > 20, done   # the solvers' traffic went through the transport
        ctx.comm.close()
    dist.barrier()
    dist.destroy_process_group()
    print(f"rank {ctx.rank} OK", flush=True)


if __name__ == "__main__":
    main()
