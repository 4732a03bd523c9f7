import json
import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
TESTS = os.path.dirname(os.path.abspath(__file__))
if TESTS not in sys.path:
    sys.path.insert(0, TESTS)

GOLDEN = os.path.join(TESTS, "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    with open(os.path.join(GOLDEN, name)) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def goldens():
    return load_golden("reference_goldens.json")


@pytest.fixture(scope="session")
def krylov_golden():
    return load_golden("krylov_cg.json")


@pytest.fixture(scope="session")
def gpu_ctx():
    """Context on cuda:0 for the -m gpu tests (single process)."""
    import torch
    if not torch.cuda.is_available():
        pytest.fail("gpu-marked test collected on a machine without a GPU")
    from legionsolvers_amd import Context
    return Context(torch.device("cuda", 0))


# ---- full-size cases: they run at the size they name or say why not -------------------------------
_FULL_SIZE_RAN = []


def require_device_memory(need_gib: float, what: str) -> None:
    """For a test of a benchmark configuration at its full size.  On a device of 192 GB or more the
    memory must be there (a failure, not a quiet downsizing); on a smaller device the test is skipped
    with the reason.  What ran is listed in the terminal summary, which `pytest -q` keeps."""
    import gc
    import torch
    gc.collect()
    torch.cuda.empty_cache()  # what earlier tests left in torch's cache is free memory too
    free, total = torch.cuda.mem_get_info()
    if total >= 192e9:
        assert free >= need_gib * (1 << 30), (
            f"{what}: needs {need_gib} GiB of free device memory, {free / 2**30:.0f} GiB of "
            f"{total / 2**30:.0f} GiB are free")
    elif free < need_gib * (1 << 30):
        pytest.skip(f"{what}: needs {need_gib} GiB of free device memory, this device has "
                    f"{free / 2**30:.0f} GiB free of {total / 2**30:.0f} GiB")
    _FULL_SIZE_RAN.append(what)


def pytest_terminal_summary(terminalreporter):
    for what in _FULL_SIZE_RAN:
        terminalreporter.write_line(f"full size: {what}")
