#!/usr/bin/env python3
"""Regenerates the golden fixtures under tests/golden/ from the reference checkout.

Run in the build container only (`python3 -B tests/golden/make_golden.py`); the GPU box has no
reference checkout and only ever reads the JSON files this script wrote.

  reference_goldens.json   the expectations the reference's own test driver holds
                           (test_all.py:19-133, 224: CG residual history as a sorted set, the
                           Test03/04 partition print-out, Test02's six zeros), read as data.
  krylov_cg.json           outputs of the reference's scripts/krylov.py (dense numpy CG) on the
                           1-D negative Laplacian, b = 1, x0 = 0: the residual-norm-squared
                           history IN ORDER (the driver only keeps a sorted set) and iterates.
"""
import ast
import json
import os
import sys

sys.dont_write_bytecode = True  # the reference mount must stay untouched

import numpy as np

REFERENCE = os.environ.get("LSK_REFERENCE", "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))


def reference_goldens():
    tree = ast.parse(open(os.path.join(REFERENCE, "test_all.py")).read())
    found = {}
    for node in tree.body:
        if isinstance(node, ast.AnnAssign) and isinstance(node.target, ast.Name):
            if node.target.id in ("EXPECTED_PARTITION_OUTPUT", "EXPECTED_CG_OUTPUT"):
                found[node.target.id] = [b.decode() for b in ast.literal_eval(node.value)]
    assert set(found) == {"EXPECTED_PARTITION_OUTPUT", "EXPECTED_CG_OUTPUT"}
    return {
        "source": "test_all.py:19-133,224",
        "expected_partition_output": found["EXPECTED_PARTITION_OUTPUT"],
        "expected_cg_output_sorted_set": found["EXPECTED_CG_OUTPUT"],
        "test02_stdout": ["0"] * 6,            # test_all.py:224
        "test01_scalar_chain": {"expr": "(2+10)/(2+2)-2", "value": 1.0},  # Test01:17-32
        "cg_problem": {"n": 100, "pieces": 4, "iterations": 10, "rhs": 1.0, "x0": 0.0},
        "partition_problem": {"n": 20, "pieces": 4},
    }


def krylov_fixture():
    sys.path.insert(0, os.path.join(REFERENCE, "scripts"))
    import krylov  # the reference's own numpy CG

    out = {"source": "scripts/krylov.py:5-40 (cg)", "cases": []}
    for n, its in ((100, 10), (16, 6)):
        A = 2.0 * np.eye(n) - np.eye(n, k=1) - np.eye(n, k=-1)
        b = np.ones(n)
        iterates = krylov.cg(A, b, np.zeros(n))
        hist = [float(b @ b)] + [float((b - A @ x) @ (b - A @ x)) for x in iterates[:its]]
        out["cases"].append({
            "n": n, "iterations": its, "residual_norm_squared": hist,
            "x_final": [float(v) for v in iterates[its - 1]],
        })
    return out


def main():
    with open(os.path.join(HERE, "reference_goldens.json"), "w") as f:
        json.dump(reference_goldens(), f, indent=1)
    with open(os.path.join(HERE, "krylov_cg.json"), "w") as f:
        json.dump(krylov_fixture(), f, indent=1)
    print("wrote reference_goldens.json, krylov_cg.json")


if __name__ == "__main__":
    main()
