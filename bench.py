#!/usr/bin/env python3
"""Headline benchmark: CG iterations/s and CSR SpMV GB/s (fp64) on BenchmarkStencil's 3-D
27-point stencil at 512^3, on N MI355X of one node.

    python bench.py --gpus 1 --steps 50 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A step is one CGSolver::step() over the whole (fixed-size) problem, which is split into N
equal row blocks, one per rank (strong scaling).  Inputs (matrix, rhs = 1, x0 = 0) are generated
on the device and resident in HBM before the timed region.  Rank 0 prints ONE JSON line.

Besides the contract's fields the line carries:
  roofline      the CSR SpMV kernel of the timed steps against the HBM roof: algorithmic bytes of
                one launch in the layout it reads (the sliced-pattern copy: padded entries * 8 +
                rows * 16 + 16 per block + 4 per row of a masked wave; else nnz*12 + rows*20 on the
                set-up-time 32-bit structure, nnz*16 + rows*32 on the arrays as handed over) / its
                mean duration from HIP events
  roofline_32bit_structure, roofline_int64_columns   the same for the kernels that read the
                32-bit copy of the structure and the reference's 64-bit column field
  cpu_baseline  the CPU oracle's CG step (OpenMP over row blocks) on the same matrix, copied to the host,
                rank 0 at N = 1 only
  transport, rccl_ranks   what carries the halo exchange and the dots' all-reduce between ranks, and how
                many ranks the RCCL communicator behind it reports (null on one rank)
  value_reference_layout, value_graph_replay, setup_s, host_ms_per_step, value_with_redraws (--redraws)   see "notes" in the line
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)


def launch_ranks_if_needed(argv) -> None:
    """`python bench.py --gpus N` with N > 1 and no launcher around it (WORLD_SIZE unset): start the N
    ranks as a CHILD job, `python -m torch.distributed.run --nproc-per-node N ... bench.py <same flags>`,
    pass its output through and exit with its code.  Runs before this process has imported torch or
    touched a GPU; the child is a fresh process, never a re-exec of this one.  When the child fails
    with the library's own RCCL transport, it is started once more with LSK_TRANSPORT=torch."""
    import socket
    import subprocess
    gpus = 1
    for i, a in enumerate(argv):
        if a == "--gpus" and i + 1 < len(argv):
            gpus = int(argv[i + 1])
        elif a.startswith("--gpus="):
            gpus = int(a.split("=", 1)[1])
    if gpus <= 1 or "WORLD_SIZE" in os.environ:
        return
    rc = 1
    attempts = [dict(os.environ)]
    if os.environ.get("LSK_TRANSPORT") in (None, "", "lsk") and os.environ.get("LSK_DIST_BACKEND", "nccl") == "nccl":
        attempts.append(dict(os.environ, LSK_TRANSPORT="torch"))
    for n, env in enumerate(attempts):
        with socket.socket() as sock:
            sock.bind(("127.0.0.1", 0))
            port = sock.getsockname()[1]
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(gpus),
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__), *argv]
        print(f"[bench] WORLD_SIZE is unset: starting {gpus} ranks: {' '.join(cmd)}"
              + (" (second attempt, LSK_TRANSPORT=torch)" if n else ""), file=sys.stderr, flush=True)
        rc = subprocess.call(cmd, env=env, cwd=REPO)
        if rc == 0:
            break
    sys.exit(rc)


if __name__ == "__main__":
    launch_ranks_if_needed(sys.argv[1:])
    # (ranks of one node share device memory handles through dmabuf on these machines; the variable is
    # exported there already, this only covers an environment that lost it)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from legionsolvers_amd import CGSolver, Context, IndexPartition, IndexSpace, PartitionedVector  # noqa: E402
from legionsolvers_amd import benchmark_stencil as bs  # noqa: E402

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak


def spmv_kernel_time(ctx, matrix, planner, reps: int, form: str) -> float:
    """Mean seconds of one CSR SpMV launch on this rank (the kernel alone, no halo traffic),
    from HIP events recorded on the launch stream.  form: "sliced" = whatever the planner launches
    (the sliced-pattern copy when the matrix got one), "c32" = the set-up-time 32-bit copy of the
    structure, "int64" = the reference's 64-bit columns and rects as handed over."""
    entry = planner.row_partitioned_matrices[0]
    src = planner.get_vector(2, 0)  # P
    dst = planner.get_vector(3, 0)  # Q
    plan = entry.plan
    x_lo, x_hi = src.own_lo, src.own_hi          # the window the planner's products use
    if plan.recvs or plan.sends:
        x_lo, x_hi = x_lo - plan.pad_lo, x_hi + plan.pad_hi
    base = src.own_lo - src.pad_lo
    x = src.data[x_lo - base:x_hi - base + 1]
    held = matrix.sliced_window, matrix.col32_window
    # the variant the timed CG steps launched: with the shares of p.Ap when the planner fuses them in
    dot_with = src if (form == "sliced" and planner_fuses_dot(planner)) else None
    try:
        if form in ("c32", "int64"):
            matrix.sliced_window = None
        if form == "int64":
            matrix.col32_window = None
        return spmv_kernel_time_inner(ctx, matrix, entry, dst, x, x_lo, x_hi, reps, dot_with)
    finally:
        matrix.sliced_window, matrix.col32_window = held


def planner_fuses_dot(planner) -> bool:
    return planner._fused_dot_used > 0 and planner.row_partitioned_matrices[0].matrix.fuses_dot()


def spmv_kernel_time_inner(ctx, matrix, entry, dst, x, x_lo, x_hi, reps: int, dot_with=None) -> float:
    for _ in range(3):
        matrix.matvec(dst, x, x_lo, x_hi, entry.kernel_partition, 0, dot_with=dot_with)
    start = torch.cuda.Event(enable_timing=True)
    stop = torch.cuda.Event(enable_timing=True)
    ctx.synchronize()
    start.record()
    for _ in range(reps):
        matrix.matvec(dst, x, x_lo, x_hi, entry.kernel_partition, 0, dot_with=dot_with)
    stop.record()
    stop.synchronize()
    return start.elapsed_time(stop) * 1e-3 / reps


def cpu_baseline(matrix, n1: int, gpu_history=None, gpu_product=None):
    """CG iterations/s of the CPU oracle on this box's host cores, on the SAME matrix: the device
    arrays (bit-identical to what the oracle's own generator builds) are copied to the host, then a few
    CG steps run with one OpenMP thread per row block, and one step on a single thread.  When the host
    lacks the memory for the whole matrix, a 256^3 sample of the same operator is timed instead and
    scaled by rows (named in `sample`).  gpu_history: r.r after each of the GPU solver's first steps (from
    the same right-hand side and starting vector); on the whole matrix the oracle's own r.r after each of
    its steps is set beside it (`parity`): the checker used as the checker, at the benchmark's size.
    gpu_product: (x, y, sliced) of one GPU product y = A x; the oracle's CSR loop repeats it on the host and the
    two are compared bit by bit (the sliced-pattern product sums every row in the CPU loop's order)."""
    import numpy as np
    import oracle
    affinity = len(os.sched_getaffinity(0))
    env_threads = os.environ.get("OMP_NUM_THREADS")
    # the box's CPU share is what OMP_NUM_THREADS says when it is set (the affinity mask of a GPU box
    # lists every CPU of the machine, of which one job gets a share)
    cores = oracle.set_threads(min(affinity, int(env_threads)) if env_threads else affinity)
    rows = matrix.rows
    need = matrix.rects.numel() * 8 + matrix.col.numel() * 8 + matrix.val.numel() * 8 + 5 * rows * 8
    try:
        import psutil
        available = psutil.virtual_memory().available
    except Exception:  # noqa: BLE001
        available = 0
    sample_n = int(os.environ.get("LSK_CPU_SAMPLE", "0"))
    t_copy = time.perf_counter()
    if not sample_n and available > 1.3 * need:
        rects, col, val = (t.cpu().numpy() for t in (matrix.rects, matrix.col, matrix._val_given))
        scale, where = 1.0, f"the benchmark's own matrix ({n1}^3: {rows} rows, {val.size} entries, copied from the device)"
    else:
        sample_n = sample_n or 256
        offs, cfs = oracle.benchmark_stencil_offsets(4)
        rects, _, col, val = oracle.linearized_stencil((sample_n,) * 3, offs, cfs)
        scale = sample_n ** 3 / rows
        rows = sample_n ** 3
        where = (f"a {sample_n}^3 sample of the same operator ({rows} rows, {val.size} entries; the host has "
                 f"{available / 2**30:.0f} GiB free, the whole matrix needs {need / 2**30:.0f}), scaled by rows to {n1}^3")
    t_copy = time.perf_counter() - t_copy
    sol, r, p, q = np.zeros(rows), np.ones(rows), np.ones(rows), np.zeros(rows)
    rr = float(rows)
    history = [rr]
    rr = oracle.cg_step_mt(rects, col, val, sol, p, q, r, rr, cores)  # warm-up (first touch of the vectors)
    history.append(float(rr))
    steps = 3
    t0 = time.perf_counter()
    for _ in range(steps):
        rr = oracle.cg_step_mt(rects, col, val, sol, p, q, r, rr, cores)
        history.append(float(rr))
    dt = (time.perf_counter() - t0) / steps
    oracle.set_threads(1)
    t0 = time.perf_counter()
    rr = oracle.cg_step_mt(rects, col, val, sol, p, q, r, rr, 1)
    history.append(float(rr))
    dt1 = time.perf_counter() - t0
    oracle.set_threads(cores)
    parity = None
    if scale == 1.0 and gpu_history is not None and len(gpu_history) >= len(history):
        worst = max(abs(g - c) / abs(c) for g, c in zip(gpu_history, history))
        parity = {"what": "r.r after each of the first CG steps, GPU solver against the CPU oracle on the same matrix, "
                          "right-hand side and starting vector (sums taken in different orders: 1e-10 is the bar)",
                  "steps": len(history) - 1, "max_rel_diff": worst, "within_1e-10": bool(worst <= 1e-10),
                  "cpu_r2": history, "gpu_r2": list(gpu_history[:len(history)])}
        if worst > 1e-10:
            print(f"[bench] PARITY: the GPU's residual history differs from the CPU oracle's by {worst:.3e}", file=sys.stderr, flush=True)
        if gpu_product is not None:
            x, y, was_sliced = gpu_product
            want = np.zeros(rows)
            t0 = time.perf_counter()
            oracle.csr_matvec(rects, int(matrix.k_base), col, val, x, 0, rows - 1, want, beta=1, mt=True)   # (rows over the threads; each row's own order)
            differing = int((want.view(np.uint64) != y.view(np.uint64)).sum())
            scale_y = float(np.abs(want).max())
            parity["product"] = {
                "what": "one product of a random vector, every row of the GPU's result against the CPU loop's, bit by bit"
                        + ("" if was_sliced else " (the general kernel adds in another order: equal to rounding, not bit by bit)"),
                "rows": int(rows), "rows_differing_in_any_bit": differing,
                "max_abs_diff_over_max_abs": float(np.abs(want - y).max()) / scale_y if scale_y else 0.0,
                "oracle_seconds": time.perf_counter() - t0}
            if was_sliced and differing:
                print(f"[bench] PARITY: {differing} rows of the sliced product differ from the CPU loop's bits", file=sys.stderr, flush=True)
    return {
        "parity": parity,
        "value": scale / dt,
        "unit": "iters/s",
        "cores": cores,
        "kind": "port",
        "sample": f"{steps} CG steps of the CPU oracle (OpenMP, {cores} threads = one row block each; the affinity mask "
                  f"lists {affinity} CPUs, OMP_NUM_THREADS={env_threads}) on {where}; {dt * 1e3:.0f} ms per step there, "
                  f"{dt1 * 1e3:.0f} ms on one thread; {t_copy:.0f} s to bring the matrix to the host",
        "value_one_thread": scale / dt1,
    }


def uniform_kernel_name(kernel: str) -> str:
    """The instance that takes the values as scalars: the wave-staged kernel with its last template
    argument (UNIFORM) true."""
    return kernel.replace(", false>", ", true>") if kernel.startswith("sliced_wave_kernel") else kernel


def wait_for_a_quiet_device(ctx, launch, limit_s: float = 8.0, need: int = 16, tolerance: float = 0.015) -> float:
    """Repeats `launch` (one product) until `need` launches in a row took within `tolerance` of the shortest
    seen, or `limit_s` seconds have passed; returns the seconds waited.  The driver scrubs the memory a
    process freed in the background at some 33 GB/s, and whatever runs meanwhile loses several per cent: a
    bench that starts right after another program of the same size (the GPU test suite, say) would otherwise
    be timed in that shadow."""
    t0 = time.perf_counter()
    best, good = float("inf"), 0
    start, stop = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    while good < need and time.perf_counter() - t0 < limit_s:
        start.record()
        launch()
        stop.record()
        stop.synchronize()
        ms = start.elapsed_time(stop)
        if ms < best * (1.0 - tolerance):
            best, good = ms, 0
        elif ms <= best * (1.0 + tolerance):
            good += 1
        else:
            good = 0
        time.sleep(0.02)
    return time.perf_counter() - t0


def library_option(key: str) -> int:
    import ctypes
    from legionsolvers_amd import _lib
    value = ctypes.c_int(0)
    _lib.call("lsk_get_option", key.encode(), ctypes.byref(value))
    return value.value


def host_ms_per_step(ctx, world: int):
    """What the host spends enqueueing one CG step: the same solver on a 32^3 grid, where the GPU
    finishes a step's kernels faster than the host issues them, so wall time per step is host time."""
    _, _, solver = bs.build_problem(ctx, 4, 32, 32, 32, world, 1, torch.float64, num_systems=1)
    eager = bs.timed_steps(ctx, solver, 20, 500) / 500 * 1e3
    if world > 1:
        return eager, None
    recorded = solver.capture()   # the same step as one HIP graph launch
    recorded.replay(20)
    ctx.synchronize()
    t0 = time.perf_counter()
    recorded.replay(500)
    ctx.synchronize()
    return eager, (time.perf_counter() - t0) / 500 * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--n", "--grid", dest="n", type=int, default=512, help="grid points per dimension")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--redraws", action="store_true",
                    help="after the timed steps, also re-make the matrix copy a few times keeping the fastest "
                         "(LSK_TUNE=2) and report the rate with it beside the headline: value_with_redraws")
    args = ap.parse_args()

    world, local_rank = bs.init_distributed()
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE is {world}: one rank per GPU "
                         "(a plain `python bench.py --gpus N` starts them itself)")
    ctx = Context(torch.device("cuda", local_rank))
    n1 = args.n
    rows = n1 ** 3
    setup_s = {}

    def stage(name, t0):
        ctx.synchronize()
        setup_s[name] = setup_s.get(name, 0.0) + time.perf_counter() - t0

    # single-system CG: the metric, with the planner's defaults (which include timing a handful of
    # workgroup placements once per matrix, and nothing that re-makes a copy or waits).
    t0 = time.perf_counter()
    _, offsets, coeffs = bs.sg.benchmark_stencil(4)
    partition = IndexPartition.equal(IndexSpace(0, rows - 1), world)
    matrix = bs.sg.create_linearized_csr_stencil_matrix(ctx, (n1, n1, n1), offsets, coeffs, partition, torch.float64)
    stage("generator", t0)
    t0 = time.perf_counter()
    _, planner, solver = bs.build_problem(ctx, 4, n1, n1, n1, world, 1, torch.float64, num_systems=1, matrix=matrix)
    stage("vectors_partitions_halo_plan", t0)
    workspace_placement = getattr(planner, "workspace_placement", None)
    for key in ("compress_32bit", "sliced_copy", "placement_tuner"):
        if key in matrix.setup_seconds:
            setup_s[key] = matrix.setup_seconds[key]
            setup_s["vectors_partitions_halo_plan"] -= matrix.setup_seconds[key]
    nnz_total = (3 * n1 - 2) ** 3
    # One untimed matvec Q <- A P before anything is timed: it opens the point-to-point
    # connections of the halo exchange, so that a run with --warmup 0 does not time their set-up.
    # CG's first step overwrites Q with the same product.
    planner.matvec(CGSolver.Q, CGSolver.P)
    ctx.comm.all_reduce_sum(torch.zeros(1, dtype=torch.float64, device=ctx.device))  # likewise the dots' all-reduce
    ctx.synchronize()
    if "placement_tuner" in matrix.setup_seconds:   # (a large piece's first product times the placements once more, on p and q)
        setup_s["placement_tuner"] = matrix.setup_seconds["placement_tuner"]
    # (one rank: the matrix's own product, not the planner's -- the planner keeps count of the dot shares its
    # products leave and nobody uses, and would stop fusing them in; more ranks: a copy of this rank's part of a
    # vector, which needs no other rank -- a GPU that just ran a smaller case of the scaling series is in the same
    # shadow -- and the ranks meet again at the barrier in front of the timed steps)
    entry0 = planner.row_partitioned_matrices[0]
    p_vec, q_vec = planner.get_vector(CGSolver.P, 0), planner.get_vector(CGSolver.Q, 0)
    if world == 1:
        probe = lambda: matrix.matvec(q_vec, p_vec._own, p_vec.own_lo, p_vec.own_hi, entry0.kernel_partition, 0)  # noqa: E731
    else:
        def probe():   # (sixteen copies a sample: long enough for a few per cent to show above the timer's jitter)
            for _ in range(16):
                ctx.backend.copy(q_vec._own, p_vec._own)
    setup_s["wait_for_a_quiet_device"] = wait_for_a_quiet_device(ctx, probe, tolerance=0.015 if world == 1 else 0.03)
    seconds = bs.timed_steps(ctx, solver, args.warmup, args.steps)
    iters_per_s = args.steps / seconds
    rr = solver.residual_norm_squared[-1].get_value()
    # (r.r after the first steps, for the CPU oracle to be set beside at the end: cpu_baseline's parity)
    gpu_history = [s.get_value() for s in solver.residual_norm_squared[:min(6, len(solver.residual_norm_squared))]]
    # the same steps replayed from a HIP graph of one step (one rank; reported beside the headline)
    graph_rate = None
    if world == 1:
        recorded = solver.capture()
        recorded.replay(max(args.warmup, 1))
        ctx.synchronize()
        t0 = time.perf_counter()
        recorded.replay(args.steps)
        ctx.synchronize()
        graph_rate = args.steps / (time.perf_counter() - t0)
        del recorded

    # the dominant kernel against its roofline (this rank's launch), priced on the bytes of the
    # layout it reads: the sliced-pattern copy (values in blocks of 512 rows, no column array), else
    # the 32-bit copy of the structure, else the 64-bit arrays as handed over
    sliced = bool(matrix.sliced)
    compressed = matrix.col32 is not None
    algo_bytes32 = matrix.nnz * 12 + matrix.rows * 20
    algo_bytes64 = matrix.nnz * 16 + matrix.rows * 32
    if sliced:
        parts = list(matrix.sliced.values())
        algo_bytes = sum(p.algorithmic_bytes(partition.pieces[c].volume, 8) for c, p in matrix.sliced.items())
        cap = 27
        dot = "true" if planner_fuses_dot(planner) else "false"
        opt = {key: library_option(key) for key in ("sliced.form", "sliced.dma", "sliced.runs")}
        if opt["sliced.form"] == 1:   # <T, CAP, DOT, K, RUN3, DMA>: wave-staged windows (the library's default)
            run3 = bool(opt["sliced.runs"]) and all(p.runs_of_three for p in parts)
            kernel = (f"sliced_wave_kernel<double, {cap}, {dot}, 1, {'true' if run3 else 'false'}, "
                      f"{'true' if opt['sliced.dma'] else 'false'}, false>")
        else:
            kernel = f"sliced_spmv_kernel<double, {cap}, {dot}, false>"
        layout = ("sliced-pattern copy made once at set-up: the values in blocks of 512 rows (16 bytes per lane and "
                  "entry), one column-pattern id per block (27 patterns), no column array and no row offsets; x "
                  "staged through LDS; every row summed in entry order, bit for bit the CPU restatement of the "
                  "reference's loop compiled without floating-point contraction")
    elif compressed:
        algo_bytes, kernel = algo_bytes32, "csr_spmv_wave_kernel<double, unsigned int, 5, false, 8, false>"
        layout = ("32-bit copies of the structure made once at set-up (window-relative columns, row offsets); "
                  "products bit-identical to those on the 64-bit arrays")
    else:
        algo_bytes, kernel = algo_bytes64, "csr_spmv_wave_kernel<double, long, 5, false, 8, false>"
        layout = "int64 global columns and 16-byte rects as handed over (the reference's layout)"
    t_spmv = spmv_kernel_time(ctx, matrix, planner, reps=20, form="sliced")
    achieved = algo_bytes / t_spmv / 1e9
    # the whole job's product: every rank's bytes over the slowest rank's launch
    job_bytes = ctx.comm.all_gather_object(algo_bytes)
    spmv_gbps_job = sum(job_bytes) / ctx.comm.max_over_ranks(t_spmv, ctx.device) / 1e9
    # the same product on the other layouts, for comparison
    made_32bit_here = False
    if sliced and not compressed:
        # (a matrix with a sliced-pattern copy gets no 32-bit copy of its structure from the planner; for this
        # comparison it is made here, for the window the planner's products use, and dropped again right after)
        entry0_ = planner.row_partitioned_matrices[0]
        own_lo, own_hi = planner.get_vector(2, 0).own_lo, planner.get_vector(2, 0).own_hi
        pad = (entry0_.plan.pad_lo, entry0_.plan.pad_hi) if (entry0_.plan.recvs or entry0_.plan.sends) else (0, 0)
        compressed = matrix.compress(own_lo - pad[0], own_hi + pad[1], entry0_.kernel_partition,
                                     planner.canonical_index_partitions[0])
        made_32bit_here = compressed
    t_spmv32 = spmv_kernel_time(ctx, matrix, planner, reps=20, form="c32") if (sliced and compressed) else t_spmv
    if made_32bit_here:
        ctx.synchronize()
        matrix.col32, matrix.col32_window, matrix.rowoff, matrix.long_rows = None, None, {}, {}
    t_spmv64 = spmv_kernel_time(ctx, matrix, planner, reps=20, form="int64") if (sliced or compressed) else t_spmv
    traffic = traffic32 = traffic64 = None
    traffic_file = os.path.join(REPO, "profiles", "spmv_hbm_traffic.json")
    traffic_source = None
    if world == 1 and n1 == 512 and os.path.exists(traffic_file):
        measured = json.load(open(traffic_file))
        traffic64 = measured.get("bytes_per_launch")
        traffic32 = measured.get("bytes_per_launch_c32")
        traffic = measured.get("bytes_per_launch_sliced") if sliced else traffic32 if compressed else traffic64
        traffic_source = ("recorded, not measured in this run: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of "
                          "tools/pmc_spmv.py, profiles/spmv_hbm_traffic.json")

    # (the optional legs below step the solver, which every rank must do or none: they go by what holds on
    # all ranks, not on this one)
    sliced_everywhere = all(ctx.comm.all_gather_object(sliced))
    uniform_everywhere = all(ctx.comm.all_gather_object(
        sliced and all(p.uniform_values for p in matrix.sliced.values())))
    with_redraws = None
    if args.redraws and sliced_everywhere:
        before = dict(matrix.setup_seconds)
        matrix.tune_sliced_copies(redraws=int(os.environ.get("LSK_TUNE_REDRAWS", matrix.SLICED_REDRAWS)))
        for key, v in matrix.setup_seconds.items():
            if v != before.get(key):
                setup_s[key + "_with_redraws"] = v - before.get(key, 0.0)
        with_redraws = args.steps / bs.timed_steps(ctx, solver, args.warmup, args.steps)

    # The same steps with the matrix values taken as scalars (option sliced.uniform): every matrix of the
    # reference's benchmark is a constant-coefficient stencil, the sliced-pattern copy finds that out bit by
    # bit at set-up, and the product then reads no value array at all.  Not the default: reported beside it.
    uniform = None
    if uniform_everywhere:
        from legionsolvers_amd import _lib
        _lib.call("lsk_set_option", b"sliced.uniform", 1)
        try:
            uni_steps = max(args.steps // 2, 2)
            uni_iters = uni_steps / bs.timed_steps(ctx, solver, 2, uni_steps)
            t_uni = spmv_kernel_time(ctx, matrix, planner, reps=20, form="sliced")
        finally:
            _lib.call("lsk_set_option", b"sliced.uniform", 0)
        uni_bytes = sum(p.algorithmic_bytes(partition.pieces[c].volume, 8, without_values=True)
                        for c, p in matrix.sliced.items())
        uniform = {"iters_per_s": uni_iters, "ms_per_launch": t_uni * 1e3, "algorithmic_bytes_per_launch": uni_bytes,
                   "achieved": uni_bytes / t_uni / 1e9, "kernel": uniform_kernel_name(kernel)}

    # the whole CG step on the reference's arrays as handed over (no set-up-time copy of any kind)
    held = matrix.sliced_window, matrix.col32_window
    matrix.sliced_window = matrix.col32_window = None
    ref_steps = max(args.steps // 4, 2)
    try:
        iters_reference_layout = ref_steps / bs.timed_steps(ctx, solver, 2, ref_steps)
    finally:
        matrix.sliced_window, matrix.col32_window = held

    del solver, planner
    # the reference's own set-up: two identical systems per step
    _, planner2, solver2 = bs.build_problem(ctx, 4, n1, n1, n1, world, 1, torch.float64, num_systems=2,
                                            matrix=matrix)
    steps2 = max(args.steps // 5, 2)
    two_system = steps2 / bs.timed_steps(ctx, solver2, 2, steps2)
    os.environ["LSK_PAIR"] = "0"   # the same steps with one product per system, as the reference launches them
    two_system_separate = steps2 / bs.timed_steps(ctx, solver2, 2, steps2)
    os.environ.pop("LSK_PAIR")
    del solver2, planner2
    # (LSK_BENCH_HOST_COST=0 leaves it out: a profile of this run then holds launches of one problem size only)
    host_ms, host_ms_graph = (host_ms_per_step(ctx, world) if os.environ.get("LSK_BENCH_HOST_COST", "1") != "0"
                              else (None, None))

    line = {
        "metric": f"CG iters/sec (fp64) on 3D 27-pt stencil {n1}^3, single system (the reference's BenchmarkStencil "
                  "steps two such systems per iteration: two_system_iters_per_s)",
        "value": iters_per_s,
        "unit": "iters/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": seconds / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "strong",
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic",
        "config": {
            "workload": f"timed layout: {layout}.  Problem: BenchmarkStencil -dim 4 -nx {n1} -ny {n1} -nz {n1} "
                        f"-solver 1 -vp {world}: 3-D 27-pt stencil, fp64, generated as CSR with int64 indices and "
                        "16-byte row rects (kept in HBM beside the copy), rhs=1, x0=0, one row block per GPU",
            "rows": rows, "nnz": nnz_total, "pieces": world,
            "structure_layout": layout,
        },
        "spmv_gbps": spmv_gbps_job,
        "two_system_iters_per_s": two_system,
        "two_system_iters_per_s_separate_products": two_system_separate,
        "value_uniform_coefficients": uniform["iters_per_s"] if uniform else None,
        "value_with_redraws": with_redraws,
        "transport": ctx.comm.transport if world > 1 else None,
        "rccl_ranks": ctx.comm.rccl_ranks,
        "transport_note": ctx.comm.transport_note,
        "setup_notes": list(getattr(matrix, "setup_notes", [])) or None,   # a set-up-time copy left out or given up for lack of memory
        "workspace_placement": workspace_placement,  # where in HBM the planner put its workspace vectors, by timing
        "value_reference_layout": iters_reference_layout,
        "setup_s": {k: round(v, 3) for k, v in setup_s.items()},
        "host_ms_per_step": host_ms,
        "host_ms_per_step_graph_replay": host_ms_graph,
        "value_graph_replay": graph_rate,
        "notes": {
            "two_system_iters_per_s": "the reference's BenchmarkStencil as it runs: two systems on one matrix per step; "
                                      "both products share one pass over the matrix values (lsk_csr_sliced_spmv2_*), each "
                                      "bit for bit what its own product gives; ..._separate_products: one product per system",
            "value_uniform_coefficients": "the same steps with option sliced.uniform = 1 (off by default): the matrix is a "
                                          "constant-coefficient stencil, which the sliced-pattern copy verifies bit by bit at "
                                          "set-up, and the product takes the 27 coefficients as scalars instead of reading "
                                          "the value array; same multiplications and additions in the same order, same bits "
                                          "(roofline_uniform_coefficients: that kernel, not bound by HBM any more)",
            "workspace_placement": "a planner default, not a bench setting: before allocating its workspace the planner times "
                                   "one product into a vector here, into one behind a block of a third of the device and "
                                   "into one behind two (HBM answers in three parts; a product is faster when it writes "
                                   "into another part than it reads the values from) and allocates where that was fastest "
                                   "(LSK_PLACE_VECTORS=0: where the allocator puts it; as the first process on a fresh "
                                   "device that is 161 against 169.6 it/s, profiles/r03_bench_five_fresh_boxes.log and "
                                   "r03_bench_v7_no_cpu_baseline.json)",
            "value_with_redraws": "only with --redraws: the same steps after the opt-in set-up-time tuning LSK_TUNE=2 (the "
                                  "copy re-made up to 6 times keeping the fastest, then a wait for the driver's scrubbing "
                                  "of freed memory); the headline `value` runs without it",
            "transport": "lsk = the library's RCCL layer (lsk_allreduce_sum_*, lsk_halo_exchange_* of include/lsk.h, halo on "
                         "a second stream of the same priority); torch = torch.distributed; rccl_ranks as lsk_comm_rank "
                         "reports it",
            "value_reference_layout": "the same steps with every product on the int64 columns and 16-byte rects as handed "
                                      "over (csr_spmv_wave_kernel<double, long, ...>), no set-up-time copy",
            "setup_s": "wall-clock seconds before the first timed step, stage by stage, this rank (wait_for_a_quiet_device: "
                       "products repeated until 16 in a row take the same time, at most 8 s -- memory freed by a program "
                       "that ran before is scrubbed in the background and slows whatever runs meanwhile)",
            "host_ms_per_step": "wall time per CG step on a 32^3 grid with the same ranks, where the host, not the GPU, "
                                "is the limit: what enqueueing one step costs; ..._graph_replay: the same step "
                                "recorded once as a HIP graph and replayed (CGSolver.capture, one rank)",
            "value_graph_replay": "the headline's steps replayed from a HIP graph of one step (one rank): the same "
                                  "kernels, one launch call per step",
        },
        "residual_norm_squared_last": rr,
        "roofline": {
            "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic, "traffic_source": traffic_source,
            "kernel": kernel,
            "ms_per_launch": t_spmv * 1e3,
            "algorithmic_bytes_per_launch": algo_bytes,
            "tile_placement": sorted(set((matrix.sliced_placement if sliced else matrix.placement).values())) or [0],
        },
        "roofline_uniform_coefficients": None if not uniform else {
            "bound": "hbm", "achieved": uniform["achieved"], "peak": HBM_PEAK_GBPS, "unit": "GB/s",
            "frac": uniform["achieved"] / HBM_PEAK_GBPS, "traffic": None, "kernel": uniform["kernel"],
            "ms_per_launch": uniform["ms_per_launch"], "algorithmic_bytes_per_launch": uniform["algorithmic_bytes_per_launch"],
        },
        "roofline_32bit_structure": {
            "bound": "hbm", "achieved": algo_bytes32 / t_spmv32 / 1e9, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
            "frac": algo_bytes32 / t_spmv32 / 1e9 / HBM_PEAK_GBPS, "traffic": traffic32,
            "kernel": "csr_spmv_wave_kernel<double, unsigned int, 5, false, 8, false>", "ms_per_launch": t_spmv32 * 1e3,
            "algorithmic_bytes_per_launch": algo_bytes32,
        },
        "roofline_int64_columns": {
            "bound": "hbm", "achieved": algo_bytes64 / t_spmv64 / 1e9, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
            "frac": algo_bytes64 / t_spmv64 / 1e9 / HBM_PEAK_GBPS, "traffic": traffic64,
            "kernel": "csr_spmv_wave_kernel<double, long, 5, false, 8, false>", "ms_per_launch": t_spmv64 * 1e3,
            "algorithmic_bytes_per_launch": algo_bytes64,
        },
    }
    if ctx.rank == 0 and world == 1 and not args.no_cpu_baseline:
        # one product of a random vector as the planner launches it, for the oracle to repeat on the host
        # (cpu_baseline's parity: the sliced-pattern product claims the CPU loop's bits)
        gen = torch.Generator(device=ctx.device).manual_seed(20250310)
        p_vec._own.copy_(torch.randn(p_vec._own.numel(), dtype=torch.float64, device=ctx.device, generator=gen))
        matrix.matvec(q_vec, p_vec._own, p_vec.own_lo, p_vec.own_hi, entry0.kernel_partition, 0)
        ctx.synchronize()
        gpu_product = (p_vec._own.cpu().numpy(), q_vec._own.cpu().numpy(), bool(matrix.sliced))
        for c in list(matrix.sliced):
            matrix.sliced[c].close()
        matrix.sliced, matrix.col32, matrix.rowoff = {}, None, {}
        torch.cuda.empty_cache()
        line["cpu_baseline"] = cpu_baseline(matrix, n1, gpu_history, gpu_product)
    if ctx.rank == 0:
        print(json.dumps(line), flush=True)
    if world > 1:
        ctx.synchronize()
        dist.barrier()
        ctx.comm.close()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
