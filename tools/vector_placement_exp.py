#!/usr/bin/env python3
"""Experiment: which of the product's two vectors (input window, output) decides its time class.
Prints, twice, the time of the sliced-pattern product for every (input, output) pair out of six
fresh 1-GB buffers and the planner's three workspace vectors.  What it showed on the headline
matrix: some buffers make every product that WRITES them 0.8 ms (14 %) slower, whatever is read --
a stable property of the buffer, not of its virtual address; on top, the whole table shifts between
5.7, 6.1 and 6.5 ms levels over time.

    python tools/vector_placement_exp.py
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from legionsolvers_amd import Context  # noqa: E402
from legionsolvers_amd import benchmark_stencil as bs  # noqa: E402

ctx = Context(torch.device("cuda", 0))
matrix, planner, solver = bs.build_problem(ctx, 4, 512, 512, 512, 1, 1, torch.float64, num_systems=1)
rows = matrix.rows
bufs = {}
for name in "pqrstu":
    bufs[name] = torch.zeros(rows, dtype=torch.float64, device="cuda")
for i in range(3):
    bufs[f"w{i}"] = planner.get_vector(2 + i, 0).owned
print(", ".join(f"{k} at {v.data_ptr():#x}" for k, v in bufs.items()), flush=True)
names = list(bufs)
for rnd in range(2):
    print(f"round {rnd}: rows = input, columns = output: " + " ".join(f"{n:>6}" for n in names), flush=True)
    for xi in names:
        line = []
        for yi in names:
            if xi == yi:
                line.append("     -")
                continue
            line.append(f"{matrix.time_piece(0, 0, bufs[xi], bufs[yi]):6.3f}")
        print(f"  {xi:>5} " + " ".join(line), flush=True)
