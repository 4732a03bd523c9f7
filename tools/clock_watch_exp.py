#!/usr/bin/env python3
"""Experiment: does the product's time level follow memory the process has just FREED?  Times the
sliced-pattern product of the headline matrix in bursts of 10 launches and prints a line whenever
the level changes by more than 1.5 %; after 8 s it allocates and frees a 30-GB buffer (back to the
driver), and again after 16 s, to see whether that alone moves the level.  rocm-smi's clocks are
printed at the start and the end (they did not move in earlier runs).

    LSK_TUNE_REDRAWS=0 python tools/clock_watch_exp.py
"""
import os
import subprocess
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from legionsolvers_amd import Context  # noqa: E402
from legionsolvers_amd import benchmark_stencil as bs  # noqa: E402


def smi():
    try:
        out = subprocess.run(["rocm-smi", "--showclocks", "--csv"], capture_output=True, text=True, timeout=20).stdout
        return " | ".join(l for l in out.splitlines() if l.strip())[:200]
    except Exception as e:  # noqa: BLE001
        return f"rocm-smi failed: {e}"


ctx = Context(torch.device("cuda", 0))
matrix, planner, solver = bs.build_problem(ctx, 4, 512, 512, 512, 1, 1, torch.float64, num_systems=1)
x, y = planner.get_vector(2, 0).owned, planner.get_vector(3, 0).owned
x.zero_()
print(smi(), flush=True)
start = time.time()
last, events = None, [8.0, 16.0]
while time.time() - start < 26:
    t = matrix.time_piece(0, 0, x, y, reps=10)
    now = time.time() - start
    if last is None or abs(t - last) > 0.015 * last:
        print(f"t = {now:5.1f} s: {t:.3f} ms per launch", flush=True)
        last = t
    if events and now > events[0]:
        events.pop(0)
        big = torch.empty(30 << 30, dtype=torch.uint8, device="cuda")
        big.zero_()
        torch.cuda.synchronize()
        del big
        torch.cuda.empty_cache()
        print(f"t = {time.time() - start:5.1f} s: allocated, wrote and freed 30 GB", flush=True)
print(f"t = {time.time() - start:5.1f} s: last level {last:.3f} ms", flush=True)
print(smi(), flush=True)
