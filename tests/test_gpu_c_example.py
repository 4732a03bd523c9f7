"""The C ABI without Python underneath: examples/cg_c_abi (plain C, gcc, include/lsk.h, -llsk; built
by __graft_entry__.build()) runs Test06CSRSolveCG's system and must print the reference's
EXPECTED_CG_OUTPUT (test_all.py:130-133), in the order the reference's scripts/krylov.py gives."""
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BINARY = os.path.join(REPO, "examples", "cg_c_abi")


@pytest.mark.parametrize("traced", [False, True])
def test_c_program_reproduces_the_reference_cg_output(goldens, krylov_golden, traced):
    """traced: the steps after the first are one recorded step replayed (lsk_stream_begin_capture /
    lsk_trace_replay), where the reference's benchmark replays a Legion trace."""
    if not os.path.exists(BINARY):
        pytest.fail("examples/cg_c_abi is not built: run __graft_entry__.build()")
    res = subprocess.run([BINARY, "100", "10"] + (["trace"] if traced else []), capture_output=True, text=True,
                         timeout=120)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = res.stdout.split()
    assert sorted(lines) == sorted(goldens["expected_cg_output_sorted_set"])
    np.testing.assert_allclose([float(v) for v in lines], krylov_golden["cases"][0]["residual_norm_squared"],
                               rtol=1e-10)


def test_c_program_larger_system():
    """N = 1 000 000, 25 steps: finite, positive history from the stand-alone binary; the same lines with
    the steps replayed from a recorded one."""
    res = subprocess.run([BINARY, "1000000", "25"], capture_output=True, text=True, timeout=120)
    assert res.returncode == 0, res.stderr[-2000:]
    hist = [float(v) for v in res.stdout.split()]
    assert len(hist) == 26 and hist[0] == 1e6 and all(np.isfinite(hist)) and min(hist) > 0
    traced = subprocess.run([BINARY, "1000000", "25", "trace"], capture_output=True, text=True, timeout=120)
    assert traced.returncode == 0, traced.stderr[-2000:]
    assert traced.stdout == res.stdout
