#!/usr/bin/env python3
"""Why the reference's CPU CSR body is not a timing baseline: src/CSRMatrixTasks.cpp:73-91 visits, for
every row, EVERY entry of the kernel piece and tests whether it lies in the row's rect -- O(rows * nnz).
Times the literal restatement (oracle_csr_matvec_literal_*) against the O(nnz) form the oracle uses
elsewhere (same products, same order) on small 27-point systems; CPU only.

    python tests/time_literal_csr_loop.py
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle  # noqa: E402

offs, cfs = oracle.benchmark_stencil_offsets(4)
for n1 in (8, 12, 16):
    rects, _, col, val = oracle.linearized_stencil((n1, n1, n1), offs, cfs)
    rows = n1 ** 3
    x = np.random.default_rng(1).standard_normal(rows)
    out = {}
    for literal in (True, False):
        y = np.zeros(rows)
        t0 = time.perf_counter()
        reps = 1 if literal else 200
        for _ in range(reps):
            oracle.csr_matvec(rects, 0, col, val, x, 0, rows - 1, y, beta=0, literal=literal)
        out[literal] = ((time.perf_counter() - t0) / reps, y.copy())
    assert np.array_equal(out[True][1], out[False][1])
    print(f"27-pt {n1}^3: rows={rows} nnz={val.size}  literal {out[True][0] * 1e3:.2f} ms   "
          f"O(nnz) {out[False][0] * 1e3:.4f} ms   ratio {out[True][0] / out[False][0]:.0f}x   (results identical)",
          flush=True)
