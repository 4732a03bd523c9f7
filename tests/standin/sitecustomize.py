"""Loaded by child interpreters of the tests that put tests/standin on PYTHONPATH: with LSK_TEST_STANDIN_RCCL=1 it
installs the stand-in for the RCCL entry points (standin_rccl.py) before the program starts, so that a product
program (bench.py) runs its N > 1 path on the library's own transport on a one-GPU machine."""
import os
import sys

if os.environ.get("LSK_TEST_STANDIN_RCCL") == "1":
    here = os.path.dirname(os.path.abspath(__file__))
    repo = os.path.dirname(os.path.dirname(here))
    for p in (repo, here):
        if p not in sys.path:
            sys.path.insert(0, p)
    import standin_rccl
    standin_rccl.install()
