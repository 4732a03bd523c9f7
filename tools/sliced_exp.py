#!/usr/bin/env python3
"""Driver of tools/exp/sliced_exp.hip: times candidate forms of the sliced-pattern product on a
synthetic all-interior 27-point matrix of n^3 rows, one form per process (forms timed side by side
in one process hide their differences), and checks each against form 0.

    python tools/sliced_exp.py --build             # hipcc, here or on the GPU box
    python tools/sliced_exp.py --forms 0,1,10,20   # spawns one child per form
"""
import argparse
import ctypes
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "exp", "sliced_exp.hip")
LIB = os.path.join(HERE, "exp", "libsliced_exp.so")


def build():
    cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off",
           "-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops", SRC, "-o", LIB]
    subprocess.run(cmd, check=True)


def child(form: int, n: int, reps: int, check: bool):
    import torch
    lib = ctypes.CDLL(LIB)
    P = lambda t: ctypes.c_void_p(t.data_ptr())
    stream = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    rows = n ** 3
    H = n * n + n + 1
    dist = sorted(dx * n * n + dy * n + dz for dx in (-1, 0, 1) for dy in (-1, 0, 1) for dz in (-1, 0, 1))
    arr = (ctypes.c_int * 32)(*dist)
    assert lib.exp_set_dist(arr, 27) == 0
    layout = 1 if 10 <= form < 20 else 0
    val = torch.empty(rows * 27, dtype=torch.float64, device="cuda")
    x = torch.empty(rows + 2 * H, dtype=torch.float64, device="cuda")
    y = torch.zeros(rows, dtype=torch.float64, device="cuda")
    assert lib.exp_init(layout, rows, P(val), x.numel(), P(x), stream()) == 0
    xs = x[H:]
    launch = lambda f, v, out: lib.exp_launch(f, rows, P(v), P(xs), P(out), stream())
    for _ in range(3):
        assert launch(form, val, y) == 0
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    times = []
    for _ in range(3):
        t0.record()
        for _ in range(reps):
            launch(form, val, y)
        t1.record()
        t1.synchronize()
        times.append(t0.elapsed_time(t1) / reps)
    nbytes = rows * 27 * 8 + (rows * 16 if form < 30 else 0)
    verdict = ""
    if check and form not in (0, 1, 8) and form < 30:
        got = y.clone()
        if layout == 1:
            del val
            val = torch.empty(rows * 27, dtype=torch.float64, device="cuda")
            lib.exp_init(0, rows, P(val), x.numel(), P(x), stream())
        launch(0, val, y)
        torch.cuda.synchronize()
        verdict = "  == form 0 bit for bit" if torch.equal(got, y) else f"  DIFFERS from form 0 in {int((got != y).sum())} rows"
    print(f"form {form:2d}  n={n}  {min(times):.3f} ms (runs {', '.join(f'{t:.3f}' for t in times)})  "
          f"{nbytes / min(times) / 1e9:.2f} TB/s on {nbytes / 1e9:.2f} GB{verdict}", flush=True)


def same_process(forms, n: int, reps: int):
    """All forms on the SAME arrays (one placement in HBM for everyone), form after form, three
    rounds of `reps` launches each; the nontemporal stream of each value array first, as a measure of
    how well that array happens to be placed."""
    import torch
    lib = ctypes.CDLL(LIB)
    P = lambda t: ctypes.c_void_p(t.data_ptr())
    stream = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    rows = n ** 3
    H = n * n + n + 1
    dist = sorted(dx * n * n + dy * n + dz for dx in (-1, 0, 1) for dy in (-1, 0, 1) for dz in (-1, 0, 1))
    assert lib.exp_set_dist((ctypes.c_int * 32)(*dist), 27) == 0
    x = torch.empty(rows + 2 * H, dtype=torch.float64, device="cuda")
    vals = {}
    for layout in sorted({1 if 10 <= f < 20 or f in (22, 23, 24, 25, 26) else 0 for f in forms}):
        vals[layout] = torch.empty(rows * 27, dtype=torch.float64, device="cuda")
        assert lib.exp_init(layout, rows, P(vals[layout]), x.numel(), P(x), stream()) == 0
    y = torch.zeros(rows, dtype=torch.float64, device="cuda")
    ref = torch.zeros(rows, dtype=torch.float64, device="cuda")
    xs = x[H:]
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

    def timed(f, v, out):
        for _ in range(2):
            assert lib.exp_launch(f, rows, P(v), P(xs), P(out), stream()) == 0
        best = []
        for _ in range(3):
            t0.record()
            for _ in range(reps):
                lib.exp_launch(f, rows, P(v), P(xs), P(out), stream())
            t1.record()
            t1.synchronize()
            best.append(t0.elapsed_time(t1) / reps)
        return best

    for layout, v in vals.items():
        print(f"layout {layout}: nontemporal stream of the values {min(timed(31, v, y)):.3f} ms", flush=True)
    if 0 in vals:
        lib.exp_launch(0, rows, P(vals[0]), P(xs), P(ref), stream())
    for f in forms:
        layout = 1 if 10 <= f < 20 or f in (22, 23, 24, 25, 26) else 0
        y.zero_()
        ts = timed(f, vals[layout], y)
        verdict = ""
        if 0 in vals and f < 30 and f not in (1, 8, 14):
            verdict = "  == form 0" if torch.equal(y, ref) else f"  DIFFERS from form 0 in {int((y != ref).sum())} rows"
        nbytes = rows * 27 * 8 + (rows * 16 if f < 30 else 0)
        print(f"form {f:2d}  {min(ts):.3f} ms ({', '.join(f'{t:.3f}' for t in ts)})  "
              f"{nbytes / min(ts) / 1e9:.2f} TB/s{verdict}", flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--build", action="store_true")
    ap.add_argument("--forms", default="0,1,7,8,2,3,5,6,10,11,12,13,20,30,31,32,33")
    ap.add_argument("--n", type=int, default=512)
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--child", type=int, default=None)
    ap.add_argument("--no-check", action="store_true")
    ap.add_argument("--same-process", action="store_true")
    args = ap.parse_args()
    if args.build:
        build()
        return
    if args.same_process:
        same_process([int(f) for f in args.forms.split(",")], args.n, args.reps)
        return
    if args.child is not None:
        child(args.child, args.n, args.reps, not args.no_check)
        return
    for f in args.forms.split(","):
        cmd = [sys.executable, os.path.abspath(__file__), "--child", f, "--n", str(args.n), "--reps", str(args.reps)]
        if args.no_check:
            cmd.append("--no-check")
        res = subprocess.run(cmd, timeout=300)
        if res.returncode != 0:
            print(f"form {f}: child exited with {res.returncode}; stopping", flush=True)
            sys.exit(res.returncode)


if __name__ == "__main__":
    main()
