"""The multi-rank path on real kernels: two ranks share the one GPU of the test machine and
exchange halos / dot partials over gloo staged through host memory (RCCL refuses two ranks on
one device, so the nccl transport itself cannot be exercised on a one-GPU box).  The plan
arithmetic, halo placement, shard generators and reductions are the ones the 8-GPU job uses."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)


def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _env(extra):
    env = dict(os.environ)
    env.update({"MASTER_ADDR": "127.0.0.1", "LSK_DIST_BACKEND": "gloo", "HSA_ENABLE_IPC_MODE_LEGACY": "0"})
    env.update(extra)
    return env


@pytest.mark.parametrize("transport,ranks", [("torch", 2), ("lsk-over-stand-in", 2), ("lsk-over-stand-in", 4),
                                             ("a-communicator-that-never-comes-up", 2)])
def test_solvers_two_ranks_one_gpu(transport, ranks):
    """transport "lsk-over-stand-in": the library's own transport (comm.LskTransport: id bootstrap over the store,
    self-check, halo exchange on its second stream ordered with events) with two ranks, above a stand-in for the
    six RCCL entry points that moves the bytes over gloo (tests/standin/standin_rccl.py): everything of the
    N > 1 path but RCCL itself.  "a-communicator-that-never-comes-up": the stand-in's lsk_comm_init_rank never
    returns; after LSK_COMM_TIMEOUT seconds every rank goes on with torch.distributed and says so."""
    port = str(_free_port())
    extra = {"LSK_TEST_STANDIN_RCCL": "1", "LSK_TRANSPORT": "lsk"} if transport != "torch" else {}
    if transport == "a-communicator-that-never-comes-up":
        extra.update({"LSK_TRANSPORT": "prefer-lsk", "LSK_COMM_TIMEOUT": "3", "LSK_TEST_STANDIN_INIT_HANGS": "1"})
    procs = [subprocess.Popen([sys.executable, "-u", os.path.join(HERE, "two_rank_worker.py")],
                              env=_env({"MASTER_PORT": port, "WORLD_SIZE": str(ranks), "RANK": str(r), "LOCAL_RANK": str(r), **extra}),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(ranks)]
    outs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=240)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append(out)
    for r, (p, out) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, f"rank {r}:\n{out[-3000:]}"
        assert f"rank {r} OK" in out


def test_bench_contract_two_ranks_one_gpu():
    """bench.py under the driver's launch line for N = 2 (small grid): one JSON line from rank 0
    with the aggregate metric, strong scaling, no cpu_baseline."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "2", "--grid", "48"]
    res = subprocess.run(cmd, env=_env({}), capture_output=True, text=True, timeout=400, cwd=REPO)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-3000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, res.stdout
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["steps"] == 4 and line["warmup"] == 2
    assert line["scaling"] == "strong" and line["unit"] == "iters/s" and line["value"] > 0
    assert line["config"]["pieces"] == 2 and "cpu_baseline" not in line
    assert set(line["roofline"]) >= {"bound", "achieved", "peak", "unit", "frac", "traffic"}
    assert line["transport"] == "torch" and line["rccl_ranks"] is None    # gloo rehearsal: RCCL cannot share a card


def test_bench_starts_its_own_ranks_two_ranks_one_gpu():
    """The same through a plain `python bench.py --gpus 2` (no launcher, WORLD_SIZE unset): bench.py starts
    the two ranks itself as a child job and passes rank 0's line through."""
    env = _env({})
    for key in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(key, None)
    cmd = [sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--grid", "40"]
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=400, cwd=REPO)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-3000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, res.stdout
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["steps"] == 3 and line["config"]["pieces"] == 2 and line["value"] > 0
    assert line["transport"] == "torch" and line["rccl_ranks"] is None    # gloo rehearsal: RCCL cannot share a card


def test_bench_two_ranks_on_the_library_transport_over_the_stand_in():
    """bench.py under the driver's launch line for N = 2 with the library's own transport above the stand-in for
    its RCCL entry points (tests/standin/: loaded through PYTHONPATH's sitecustomize): the line reports
    transport "lsk" and the two ranks the communicator counts."""
    env = _env({"LSK_TEST_STANDIN_RCCL": "1", "LSK_TRANSPORT": "lsk"})
    env["PYTHONPATH"] = os.path.join(HERE, "standin") + os.pathsep + env.get("PYTHONPATH", "")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "2", "--grid", "48"]
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=400, cwd=REPO)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-3000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, res.stdout
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["value"] > 0 and line["config"]["pieces"] == 2
    assert line["transport"] == "lsk" and line["rccl_ranks"] == 2 and line["transport_note"] is None
