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
